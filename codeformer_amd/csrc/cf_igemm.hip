// Implicit-GEMM convolution / linear on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// One kernel template covers every dense contraction of the CodeFormer hot path
// (reference call sites: include/codeformer_hip.h, cf_conv2d):
//   out[pixel][n] = epilogue( sum_{tap, c} prologue(in[pixel + tap][c]) * W[n][c][tap] + bias[n] )
// GEMM view: M = batch*hout*wout output pixels, N = cout, K = taps*cin.
//
// Data layout in HBM: activations channels-last fp32 ([b][h][w][c]); weights pre-packed as
// [tap][cin/16][cout_pad][16] so a (tap, k-slab, n-tile) weight slab is one contiguous block.
//
// Work decomposition: a 256-thread workgroup (4 waves) owns a BM x BN output tile; BM is a TH x 16
// spatial patch of ONE image.  For each 16-channel slab the (TH*s+2) x (16*s+2) input halo patch is
// gathered once into LDS -- with the GroupNorm-apply/swish (or LeakyReLU) prologue, the nearest-x2
// upsample, the channel concat of two inputs and the zero padding all resolved in that gather -- and
// is then reused by all 9 taps; the per-tap weight slab is double-buffered in LDS and prefetched
// through registers while the MFMAs of the previous tap run.  Each wave accumulates a
// (MI*32) x (NI*32) sub-tile in MI*NI*16 accumulator registers.
#include <type_traits>

#include "cf_common.h"

// tuning knobs explored with tools/ab_variants.sh (defaults = the shipped configuration)
#ifndef CF_LOADA_TAP
#define CF_LOADA_TAP 0      // tap at which the next halo patch is prefetched into registers (0..8); 0 = a whole slab of cover (+1 % vs 8)
#endif
#ifndef CF_WAVES_PER_SIMD
#define CF_WAVES_PER_SIMD 3  // __launch_bounds__ occupancy target
#endif
#ifndef CF_EPI_WAVESYNC
#define CF_EPI_WAVESYNC 1    // 1: the epilogue transpose is wave-private (no workgroup barriers, operand loads issued first)
#endif
#ifndef CF_XCD_SWIZZLE
#define CF_XCD_SWIZZLE 1     // 1: remap workgroup ids so each XCD (private L2) owns a contiguous run of tiles (shared halos / weights)
#endif
#ifndef CF_INTERLEAVE
#define CF_INTERLEAVE 1      // 1: MFMA-first weave of fetches / LDS traffic into the MFMA stream (sched_group_barrier)
#endif
#ifndef CF_FAST_SWISH
#define CF_FAST_SWISH 1      // 1: fp32 prologue swish on v_exp_f32 / v_rcp_f32 (~3 ulp of x*sigmoid(x)) instead of expf + IEEE divide: +1.2..2.5 % on the GN-swish convs
#endif

namespace {

struct ConvArgs {
  const float* in0;
  const float* in1;
  int c0, c1, cin, nchunks;
  int batch, hin, win, hout, wout;
  int cout, cout_pad;
  int upsample, out_nchw, prologue, epilogue;
  const float* pro_scale;
  const float* pro_shift;
  const float* weight;
  const float* bias;
  const float* res;
  const float* sft_scale;
  float sft_w;
  float* out;
  double* stats_out;  // optional [batch][cout/stats_cpg][nparts][2] partial (sum, sumsq) of the OUTPUT
  int stats_cpg, nparts;
  int tiles_x, tiles_per_img, ntn;
};
// The general (EXT) instantiations take a longer argument block; the CodeFormer kernels keep the exact ConvArgs layout
// (the kernarg size feeds register allocation: growing it perturbed every instantiation's code).
struct ConvArgsExt : ConvArgs {
  int ld0, ld1, ldo;  // channel strides of in0 / in1 / out (and res, res2)
  int pad_mode;       // CF_PAD_ZERO / CF_PAD_REFLECT (3x3) / CF_PAD_EDGE (folded upsample)
  int pad_lo;         // stride 2: rows / columns of padding on the top / left (0 = CodeFormer Downsample, 1 = symmetric)
  int nt_out;         // non-temporal output stores (the first conv; cf_common.h: cf_store16)
};
// border handling of the gather for the EXT instantiations: coordinate remap instead of zero fill
__device__ __forceinline__ int cf_border(int i, int n, int mode) {
  if (mode == CF_PAD_REFLECT) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
  }
  return i < 0 ? 0 : (i >= n ? n - 1 : i);  // (also keeps the overhang of masked edge tiles addressable)
}
// Split-K instantiations (1x1 / Linear on small token matrices): workspace, ticket counters and the split count ride behind the
// ConvArgs fields, again in their own struct so the other instantiations keep their kernarg layout.
struct ConvArgsSK : ConvArgs {
  float* ws;
  unsigned* counters;
  int nsplit;
};
template <bool EXT, bool SK = false>
using ArgsOf = std::conditional_t<SK, ConvArgsSK, std::conditional_t<EXT, ConvArgsExt, ConvArgs>>;
template <bool EXT, bool SK = false>
__device__ __forceinline__ int ext_ld(const ArgsOf<EXT, SK>& a, int which, int dense) {
  if constexpr (EXT) return which == 0 ? a.ld0 : (which == 1 ? a.ld1 : a.ldo);
  return dense;
}

template <int TAPS, int STRIDE, int WM, int WN, int MI, int NI>
struct Cfg {
  static constexpr int BM = WM * MI * 32;
  static constexpr int BN = WN * NI * 32;
  static constexpr int TW = 16;
  static constexpr int TH = BM / 16;
  // TAPS: 9 = 3x3 ; 4 = nearest-x2 + 3x3 folded into four 2x2 sub-pixel convolutions (one output parity class per
  // workgroup, taps pre-summed at pack time: 4 instead of 9 MACs per weight) ; 1 = 1x1 / Linear
  static constexpr int HH = (TAPS == 9) ? (TH - 1) * STRIDE + 3 : (TAPS == 4 ? TH + 1 : TH);   // halo patch rows
  static constexpr int HWD = (TAPS == 9) ? (TW - 1) * STRIDE + 3 : (TAPS == 4 ? TW + 1 : TW);  // halo patch cols
  static constexpr int NPIX = (TAPS > 1) ? HH * HWD : BM;
  static constexpr int ABUF = (TAPS == 1) ? 3 : 1;    // 1x1: the A rows ride in a ring like the weights
  static constexpr int APT = (NPIX * 4 + 255) / 256;  // float4 gather items per thread
  static constexpr int BPT = (BN * 4 + 255) / 256;    // float4 weight items per thread
  static constexpr int BBUF = 3;                      // weight-slab ring depth in LDS
  static constexpr int LDS_MAIN = ABUF * NPIX * CF_LDK + BBUF * BN * CF_LDK;
  static constexpr int LDS_EPI = 4 * 32 * (NI * 32 + 4);  // per-wave 32-row transpose buffers of the epilogue
  static constexpr int LDS_FLOATS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
};

// x * sigmoid(x) with sigmoid = 1/(1+exp(-x)), the operation order of vqgan_arch.py:18-20
__device__ __forceinline__ float swishf(float y) { return y * (1.0f / (1.0f + expf(-y))); }

// BF16 = true: the same schedule on v_mfma_f32_32x32x16_bf16 -- a K slab is 32 channels stored as bf16 (the LDS row is
// still 64 B + 16 B pad, so every LDS address below is unchanged), activations are rounded to bf16 (RNE) in the gather
// after the prologue, weights are pre-packed bf16, accumulation stays fp32.  Used for the generator / CFT convs of the
// bf16 configurations only; the fp32 instantiations are bit-for-bit what they were.
//
// EXT = true: the general variant used by dense-block networks (Real-ESRGAN RRDBNet): inputs / outputs are channel slices
// of wider NHWC buffers (ld0 / ld1 / ldo), image sizes need not be tile multiples (edge tiles are masked in the epilogue;
// the gather already zero-fills outside the image), and the epilogue set is {none, residual, leaky, axpy, axpy2}.  Kept
// behind a template flag so the CodeFormer instantiations stay instruction-for-instruction what they were.
// F16 = true: as BF16 but IEEE half operands on v_mfma_f32_32x32x16_f16 (3 more mantissa bits, same MFMA rate) -- the operand
// format of the reference's `half=True` Real-ESRGAN (inference_codeformer.py:23-27,44) and of CodeFormer's precision='fp16';
// fp32 accumulation and fp32 tensors in HBM.
// SK = true (1x1 / Linear only): `nsplit` workgroups share an output tile, each contracting a contiguous range of K slabs; the partial
// accumulators meet in a workspace and the last arriver adds them in split order (cf_splitk_combine) before the usual epilogue.
// BIO = true (round 6, cf_conv_desc.io_bf16): in0, in1, res, sft_scale and out are bf16 tensors (2 bytes per element); the gather widens on
// load, the vector epilogue rounds once (RNE) after the GroupNorm partials were taken.  Only the instantiations the bf16 mode runs from
// 64x64 pixels up exist (direct bf16 3x3 with 64 output channels, folded upsample, fp32 1x1 on images); every other one is untouched.
template <int TAPS, int STRIDE, int WM, int WN, int MI, int NI, bool IN_NCHW, bool BF16 = false, bool EXT = false, bool F16 = false,
          bool SK = false, bool BIO = false>
__global__ __launch_bounds__(256, CF_WAVES_PER_SIMD) void igemm_kernel(const ArgsOf<EXT, SK> a) {
  using C = Cfg<TAPS, STRIDE, WM, WN, MI, NI>;
  static_assert(!BIO || (!IN_NCHW && !EXT && !SK && !F16 && STRIDE == 1), "bf16 tensors: dense NHWC stride-1 instantiations only");
  static_assert(!SK || (TAPS == 1 && !EXT && !BF16 && !F16), "split-K: 1x1 / Linear fp32 instantiations");
  constexpr bool LP = BF16 || F16;  // 16-bit MFMA operands
  static_assert(!LP || (TAPS > 1 && STRIDE == 1 && !IN_NCHW), "16-bit operand path: 3x3 stride 1 NHWC only");
  static_assert(!(BF16 && F16), "one operand format");
  static_assert(!EXT || (TAPS > 1 && !IN_NCHW && !BF16 && CF_EPI_WAVESYNC), "EXT: 3x3 / folded 2x2, NHWC, fp32 or f16");
  constexpr int KC = LP ? 32 : CF_BK;  // channels per K slab
  constexpr int AV = LP ? 2 : 1;       // float4 fetched per gather item (8 / 4 channels)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As = smem;
  float* const Bs = smem + C::ABUF * C::NPIX * CF_LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // Workgroup b runs on XCD b % 8 (observed dispatch order; speed only, never correctness).  Give every XCD a contiguous
  // run of tile ids so neighbouring tiles -- which share halo rows/columns and the weight slabs -- hit the same L2.
  int bid = blockIdx.x;
#if CF_XCD_SWIZZLE
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;  // bijective for any grid size
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
#endif
  [[maybe_unused]] int split = 0, sk_tile = 0;
  if constexpr (SK) {
    split = bid % a.nsplit;  // the splits of one tile are neighbours in the tile order
    bid /= a.nsplit;
    sk_tile = bid;
  }
  const int nt = bid % a.ntn;
  const int mt = bid / a.ntn;
  const int n0 = nt * C::BN;
  int b, y0 = 0, x0 = 0, m0 = 0;
  int sub_y = 0, sub_x = 0;  // TAPS == 4: output parity class (oy & 1, ox & 1) of this workgroup; y0/x0 are SOURCE coords
  if (TAPS > 1) {
    b = mt / a.tiles_per_img;
    int r = mt - b * a.tiles_per_img;
    if (TAPS == 4) {
      const int cls = r & 3;  // class-minor: the four parity classes of a source tile run side by side on one XCD and share its halo patch in L2 (cf_split.hip)
      r >>= 2;
      sub_y = cls >> 1;
      sub_x = cls & 1;
    }
    const int ty = r / a.tiles_x;
    y0 = ty * C::TH;
    x0 = (r - ty * a.tiles_x) * C::TW;
  } else {
    m0 = mt * C::BM;
    b = m0 / (a.hout * a.wout);
  }

  // ---- gather geometry: item j of this thread is float4 #k4 of halo pixel p = (tid>>2) + 64*j ----
  const int k4 = tid & 3;
  int pix[C::APT];  // source pixel index (b*hin+sy)*win+sx (NCHW: sy*win+sx), -1 = zero padding / unused
#pragma unroll
  for (int j = 0; j < C::APT; ++j) {
    const int p = (tid >> 2) + 64 * j;
    int v = -1;
    if (p < C::NPIX) {
      if (TAPS == 4) {
        // parity 0 reads source rows (y-1, y), parity 1 reads (y, y+1): the 2x2 footprint of the folded taps
        const int hy = p / C::HWD;
        const int hx = p - hy * C::HWD;
        int iy = y0 - 1 + sub_y + hy;
        int ix = x0 - 1 + sub_x + hx;
        if constexpr (EXT) {
          if (a.pad_mode != CF_PAD_ZERO) {
            iy = cf_border(iy, a.hin, a.pad_mode);
            ix = cf_border(ix, a.win, a.pad_mode);
          }
        }
        if (iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) v = (b * a.hin + iy) * a.win + ix;
      } else if (TAPS == 9) {
        const int hy = p / C::HWD;
        const int hx = p - hy * C::HWD;
        int pad = (STRIDE == 1) ? 1 : 0;
        if constexpr (EXT && STRIDE == 2) pad = a.pad_lo;
        int iy = y0 * STRIDE - pad + hy;
        int ix = x0 * STRIDE - pad + hx;
        if constexpr (EXT) {
          if (a.pad_mode != CF_PAD_ZERO) {
            iy = cf_border(iy, a.hin, a.pad_mode);
            ix = cf_border(ix, a.win, a.pad_mode);
          }
        }
        if (iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) v = IN_NCHW ? (iy * a.win + ix) : ((b * a.hin + iy) * a.win + ix);
      } else {
        v = m0 + p;
      }
    }
    pix[j] = v;
  }

  // Loads are issued UNCONDITIONALLY from a clamped (always valid) address and zeroed by a select afterwards:
  // a load under a divergent branch makes hipcc wait for it on the spot, which would serialise the prefetch.
  auto load_A = [&](int chunk, f32x4(&ra)[C::APT * AV]) {
    const int c = chunk * KC + k4 * (KC / 4);
    if (IN_NCHW) {
      const size_t plane = (size_t)a.hin * a.win;
      const float* base = a.in0 + (size_t)b * a.c0 * plane;
      const int c1 = a.c0 > 1 ? 1 : 0, c2 = a.c0 > 2 ? 2 : 0, c3 = a.c0 > 3 ? 3 : 0;
#pragma unroll
      for (int j = 0; j < C::APT; ++j) {
        const int pj = pix[j] < 0 ? 0 : pix[j];
        f32x4 v;
        v[0] = base[pj];
        v[1] = base[c1 * plane + pj];
        v[2] = base[c2 * plane + pj];
        v[3] = base[c3 * plane + pj];
        const bool ok = pix[j] >= 0 && c == 0;
        v[0] = ok ? v[0] : 0.f;
        v[1] = (ok && a.c0 > 1) ? v[1] : 0.f;
        v[2] = (ok && a.c0 > 2) ? v[2] : 0.f;
        v[3] = (ok && a.c0 > 3) ? v[3] : 0.f;
        ra[j * AV] = v;
      }
    } else {
      const bool first = c < a.c0;
      const float* src = first ? a.in0 : a.in1;
      int cs = first ? a.c0 : a.c1;
      if constexpr (EXT) cs = first ? a.ld0 : a.ld1;
      const int cc = first ? c : c - a.c0;
#pragma unroll
      for (int j = 0; j < C::APT; ++j) {
        const int pj = pix[j] < 0 ? 0 : pix[j];
        if constexpr (BIO) {  // bf16 tensor: this item's KC / 4 channels are 8 (fp32 operands) or 16 (16-bit operands) bytes
          const unsigned short* sp = reinterpret_cast<const unsigned short*>(src) + (size_t)pj * cs + cc;
          if constexpr (AV == 2) {
            const cf_u32x4 q = *reinterpret_cast<const cf_u32x4*>(sp);
            ra[j * 2] = cf_bf16x4_widen(cf_u32x2{q[0], q[1]});
            ra[j * 2 + 1] = cf_bf16x4_widen(cf_u32x2{q[2], q[3]});
          } else {
            ra[j] = cf_bf16x4_widen(*reinterpret_cast<const cf_u32x2*>(sp));
          }
#pragma unroll
          for (int u = 0; u < AV; ++u)
            if (pix[j] < 0) ra[j * AV + u] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int u = 0; u < AV; ++u) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)pj * cs + cc + 4 * u);
            if (pix[j] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
            ra[j * AV + u] = v;
          }
        }
      }
    }
  };

  // zero padding must stay exactly zero: it pads the conv INPUT, i.e. the post-activation tensor
  auto store_A_mode = [&](int buf, const f32x4(&ra)[C::APT * AV], int chunk, auto mode) {
    constexpr int PRO = decltype(mode)::value;
    f32x4 sc[AV], sh[AV];
#pragma unroll
    for (int u = 0; u < AV; ++u) {
      sc[u] = f32x4{1.f, 1.f, 1.f, 1.f};
      sh[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (PRO == CF_PRO_AFFINE || PRO == CF_PRO_AFFINE_SWISH) {
        const int c = chunk * KC + k4 * (KC / 4) + 4 * u;
        sc[u] = *reinterpret_cast<const f32x4*>(a.pro_scale + (size_t)b * a.cin + c);
        sh[u] = *reinterpret_cast<const f32x4*>(a.pro_shift + (size_t)b * a.cin + c);
      }
    }
    float* dst = As + buf * (C::NPIX * CF_LDK) + k4 * 4;
#pragma unroll
    for (int j = 0; j < C::APT; ++j) {
      const int p = (tid >> 2) + 64 * j;
      if ((j + 1) * 64 <= C::NPIX || p < C::NPIX) {
        const bool valid = pix[j] >= 0;
        f32x4 v[AV];
#pragma unroll
        for (int u = 0; u < AV; ++u) {
          v[u] = ra[j * AV + u];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y = v[u][e];
            if (PRO == CF_PRO_AFFINE) y = y * sc[u][e] + sh[u][e];
            if (PRO == CF_PRO_AFFINE_SWISH) {
              y = y * sc[u][e] + sh[u][e];
              y = (LP || CF_FAST_SWISH) ? y * __frcp_rn(1.0f + __expf(-y)) : swishf(y);  // bf16 operands: fast exp/rcp are far below the rounding
            }
            if (PRO == CF_PRO_LEAKY) y = y > 0.f ? y : 0.2f * y;
            v[u][e] = valid ? y : 0.f;
          }
        }
        if constexpr (F16) {
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          f32x4 packed;  // 8 channels -> 8 IEEE halves (v_cvt_f16_f32, round-to-nearest-even) in 16 bytes
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const float flo = v[h >> 1][(h & 1) * 2], fhi = v[h >> 1][(h & 1) * 2 + 1];
            const f16x2 pr = {(_Float16)flo, (_Float16)fhi};
            packed[h] = __builtin_bit_cast(float, pr);
          }
          *reinterpret_cast<f32x4*>(dst + p * CF_LDK) = packed;
        } else if constexpr (BF16) {
          f32x4 packed;  // 8 channels -> 8 bf16 (round-to-nearest-even) in 16 bytes, channel order preserved
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            // (copy the lanes out first: __builtin_bit_cast on a vector-element lvalue reads element 0)
            const float flo = v[h >> 1][(h & 1) * 2], fhi = v[h >> 1][(h & 1) * 2 + 1];
            unsigned lo = __builtin_bit_cast(unsigned, flo);
            unsigned hi = __builtin_bit_cast(unsigned, fhi);
            lo += 0x7fffu + ((lo >> 16) & 1u);
            hi += 0x7fffu + ((hi >> 16) & 1u);
            packed[h] = __builtin_bit_cast(float, (lo >> 16) | (hi & 0xffff0000u));
          }
          *reinterpret_cast<f32x4*>(dst + p * CF_LDK) = packed;
        } else {
          *reinterpret_cast<f32x4*>(dst + p * CF_LDK) = v[0];
        }
      }
    }
  };
  auto store_A = [&](int buf, const f32x4(&ra)[C::APT * AV], int chunk) {
    switch (a.prologue) {
      case CF_PRO_AFFINE: store_A_mode(buf, ra, chunk, std::integral_constant<int, CF_PRO_AFFINE>{}); break;
      case CF_PRO_AFFINE_SWISH: store_A_mode(buf, ra, chunk, std::integral_constant<int, CF_PRO_AFFINE_SWISH>{}); break;
      case CF_PRO_LEAKY: store_A_mode(buf, ra, chunk, std::integral_constant<int, CF_PRO_LEAKY>{}); break;
      default: store_A_mode(buf, ra, chunk, std::integral_constant<int, CF_PRO_NONE>{}); break;
    }
  };

  constexpr bool B_FULL = (C::BN * 4) % 256 == 0;  // every thread has BPT items (no tail guard)
  auto load_B = [&](int step, f32x4(&rb)[C::BPT]) {
    const int chunk = step / TAPS;
    const int tap = step - chunk * TAPS;
    const int cls_tap = (TAPS == 4) ? (sub_y * 2 + sub_x) * 4 + tap : tap;  // sub-pixel: [class][tap] slabs
    const float* src = a.weight + ((size_t)(cls_tap * a.nchunks + chunk) * a.cout_pad + n0) * CF_BK;
#pragma unroll
    for (int j = 0; j < C::BPT; ++j) {
      int f = tid + 256 * j;
      if (!B_FULL) f = f < C::BN * 4 ? f : 0;
      rb[j] = *reinterpret_cast<const f32x4*>(src + f * 4);
    }
  };
  auto store_B = [&](int buf, const f32x4(&rb)[C::BPT]) {
    float* dst = Bs + buf * (C::BN * CF_LDK);
#pragma unroll
    for (int j = 0; j < C::BPT; ++j) {
      const int f = tid + 256 * j;
      if (B_FULL || f < C::BN * 4) *reinterpret_cast<f32x4*>(dst + (f >> 2) * CF_LDK + (f & 3) * 4) = rb[j];
    }
  };

  // ---- MFMA operand rows of this lane ----
  int a_off[MI], b_off[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * (MI * 32) + mi * 32 + l31;
    if (TAPS > 1) {
      const int py = row >> 4, px = row & 15;
      a_off[mi] = ((py * STRIDE) * C::HWD + px * STRIDE) * CF_LDK + half * 4;
    } else {
      a_off[mi] = row * CF_LDK + half * 4;
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_off[ni] = (wn * (NI * 32) + ni * 32 + l31) * CF_LDK + half * 4;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nsteps = a.nchunks * TAPS;
  f32x4 ra[C::APT * AV];
  f32x4 rb[C::BPT];

  if constexpr (TAPS > 1) {
    // ---- software-pipelined schedule (3x3 / folded 2x2) -----------------------------------------------------------------
    // Weight slabs live in a 3-deep LDS ring: slab t is fetched from HBM/L2 at the start of step t-2, written to
    // LDS at the end of step t-2, made visible by the barrier that opens step t-1, and its first fragments are read
    // in the MIDDLE of step t-1 -- so no wave ever waits on an LDS round trip between two MFMA blocks:
    //     step s:  [barrier] issue loads B(s+2) | read frags(s, k 8..15) | 16 MFMA on frags(s, k 0..7)
    //                        read frags(s+1, k 0..7) | 16 MFMA on frags(s, k 8..15) | LDS write B(s+2)
    // RAW: B(s+1) was written before the barrier opening step s.  WAR: ring slot (s+2)%3 last held B(s-1), whose
    // reads every wave finished before arriving at the barrier opening step s.  The halo patch is single-buffered,
    // so the slab boundary (1 step in 9) drains: barrier, patch write, barrier, fragment read.
    auto read_frags = [&](f32x4(&af)[MI], f32x4(&bf)[NI], int tapoff, int bslot, int kg) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(As + a_off[mi] + tapoff + kg * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const f32x4*>(Bs + bslot * (C::BN * CF_LDK) + b_off[ni] + kg * 8);
    };
    auto mma16 = [&](const f32x4(&af)[MI], const f32x4(&bf)[NI]) {
      if constexpr (F16) {
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[mi]),
                                                                 __builtin_bit_cast(f16x8, bf[ni]), acc[mi][ni], 0, 0, 0);
      } else if constexpr (BF16) {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mi]),
                                                                  __builtin_bit_cast(bf16x8, bf[ni]), acc[mi][ni], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
      }
    };
    auto tap_off = [](int tap) { return (TAPS == 4 ? (tap >> 1) * C::HWD + (tap & 1) : (tap / 3) * C::HWD + (tap % 3)) * CF_LDK; };

    {
      f32x4 rb1[C::BPT];  // all three prologue fetches in flight together: one exposed HBM/L2 latency, not two
      load_A(0, ra);
      load_B(0, rb);
      load_B(1 < nsteps ? 1 : 0, rb1);
      store_A(0, ra, 0);
      store_B(0, rb);
      store_B(1, rb1);
    }
    __syncthreads();
    f32x4 ax[MI], bx[NI], ay[MI], by[NI];
    read_frags(ax, bx, tap_off(0), 0, 0);
    int slot = 0;  // ring slot of the current step's weight slab
    int step = 0;
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap, ++step) {
        const int slot1 = slot == 2 ? 0 : slot + 1;
        const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
        constexpr int NM = LP ? MI * NI : MI * NI * 4;                      // MFMAs per half step
        constexpr int NA = (CF_LOADA_TAP >= 0) ? C::APT * AV : 0;           // halo-patch fetches riding in one step
        constexpr bool WEAVE = CF_INTERLEAVE && !LP && NM >= MI + NI + C::BPT + NA;
        // ---- first half: fetch slab s+2 (and, once per slab, the next halo patch), read frags(s, k 8..15), MFMA on frags(s, k 0..7)
        load_B(step + 2 < nsteps ? step + 2 : nsteps - 1, rb);  // clamped: the tail prefetches are harmless re-reads
        if (tap == CF_LOADA_TAP) load_A(chunk + 1 < a.nchunks ? chunk + 1 : chunk, ra);
        read_frags(ay, by, tap_off(tap), slot, 1);
        if (!WEAVE) __builtin_amdgcn_sched_barrier(0);  // pin the fetches above the MFMA block (hipcc would sink them)
        mma16(ax, bx);
        if (WEAVE) {
          // MFMA-first weave: frags(s, k 0..7) are already in registers, so the block opens with an MFMA right behind the
          // barrier and every fetch / LDS read issues in the shadow of a 64-cycle MFMA instead of in front of the block.
#pragma unroll
          for (int i = 0; i < MI + NI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
          }
#pragma unroll
          for (int i = 0; i < C::BPT + (tap == CF_LOADA_TAP ? NA : 0); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
          }
#pragma unroll
          for (int i = 0; i < NM - (MI + NI) - C::BPT - (tap == CF_LOADA_TAP ? NA : 0); ++i)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- second half: read frags(s+1, k 0..7), MFMA on frags(s, k 8..15), LDS write of slab s+2
        if (tap != TAPS - 1) read_frags(ax, bx, tap_off(tap + 1), slot1, 0);
        if (!WEAVE) __builtin_amdgcn_sched_barrier(0);
        mma16(ay, by);
        if (!WEAVE) __builtin_amdgcn_sched_barrier(0);
        store_B(slot2, rb);
        if (WEAVE) {
#pragma unroll
          for (int i = 0; i < (tap != TAPS - 1 ? MI + NI : 0); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#pragma unroll
          for (int i = 0; i < NM - (tap != TAPS - 1 ? MI + NI : 0) - C::BPT; ++i)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
          for (int i = 0; i < C::BPT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write (waits for the slab fetched in the first half)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (tap == TAPS - 1 && chunk + 1 < a.nchunks) {
          store_A(0, ra, chunk + 1);
          __syncthreads();
          read_frags(ax, bx, tap_off(0), slot1, 0);
        }
        slot = slot1;
      }
    }
  } else {
    // ---- 1x1 / Linear: the same ring + weave schedule with BOTH operands in 3-deep LDS rings -----------------------------
    // Slab t (A rows and weight rows) is fetched at the start of step t-2, written to LDS at the end of step t-2 (the
    // GroupNorm-apply prologue of the AttnBlock q|k|v GEMM runs in that write), made visible by the barrier opening step
    // t-1 and first read in the middle of step t-1; ring slot (t+3)%3 is rewritten only after the barrier that follows
    // every wave's last read of slab t.
    auto read_frags1 = [&](f32x4(&af)[MI], f32x4(&bf)[NI], int slot_, int kg) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const f32x4*>(As + slot_ * (C::NPIX * CF_LDK) + a_off[mi] + kg * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const f32x4*>(Bs + slot_ * (C::BN * CF_LDK) + b_off[ni] + kg * 8);
    };
    auto mma16 = [&](const f32x4(&af)[MI], const f32x4(&bf)[NI]) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    };
    int n = a.nchunks, kb = 0;  // this workgroup's K slabs: [kb, kb + n)
    [[maybe_unused]] f32x16 tot[MI][NI];  // SK: ordered sum of the finished virtual chunks (nsplit == 1)
    if constexpr (SK) {
      const int per = (a.nchunks / CF_SK_SLABS / a.nsplit) * CF_SK_SLABS;  // slabs of this workgroup: whole virtual chunks (host-checked)
      kb = split * per;
      n = per;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[mi][ni][r] = 0.f;
    }
    {
      f32x4 ra1[C::APT * AV];
      f32x4 rb1[C::BPT];
      load_A(kb, ra);
      load_B(kb, rb);
      load_A(kb + (1 < n ? 1 : 0), ra1);
      load_B(kb + (1 < n ? 1 : 0), rb1);
      store_A(0, ra, kb);
      store_B(0, rb);
      store_A(1, ra1, kb + (1 < n ? 1 : 0));
      store_B(1, rb1);
    }
    __syncthreads();
    f32x4 ax[MI], bx[NI], ay[MI], by[NI];
    read_frags1(ax, bx, 0, 0);
    int slot = 0;
    for (int chunk = 0; chunk < n; ++chunk) {
      const int slot1 = slot == 2 ? 0 : slot + 1;
      const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
      const int nxt = kb + (chunk + 2 < n ? chunk + 2 : n - 1);  // clamped: the tail prefetches are harmless re-reads
      constexpr int NM = MI * NI * 4;
      constexpr bool WEAVE = CF_INTERLEAVE && NM >= MI + NI + C::BPT + C::APT * AV;
      load_B(nxt, rb);
      load_A(nxt, ra);
      read_frags1(ay, by, slot, 1);
      if (!WEAVE) __builtin_amdgcn_sched_barrier(0);
      mma16(ax, bx);
      if (WEAVE) {
#pragma unroll
        for (int i = 0; i < MI + NI; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < C::BPT + C::APT * AV; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < NM - (MI + NI) - C::BPT - C::APT * AV; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      read_frags1(ax, bx, slot1, 0);  // slab chunk+1 (on the last step: a harmless read of the clamped duplicate)
      mma16(ay, by);
      if (WEAVE) {
#pragma unroll
        for (int i = 0; i < MI + NI; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < NM - (MI + NI); ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_A(slot2, ra, nxt);
      store_B(slot2, rb);
      if constexpr (SK) {
        if ((chunk + 1) % CF_SK_SLABS == 0) {  // a virtual chunk is complete: fold it (one workgroup per tile) or park it (split)
          if (a.nsplit == 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                tot[mi][ni] += acc[mi][ni];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
              }
          } else {
            f32x4 v[MI * NI * 4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  v[(mi * NI + ni) * 4 + q] = f32x4{acc[mi][ni][q * 4], acc[mi][ni][q * 4 + 1], acc[mi][ni][q * 4 + 2], acc[mi][ni][q * 4 + 3]};
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc[mi][ni][q * 4 + e] = 0.f;
                }
            cf_splitk_park(v, a.ws, sk_tile, (kb + chunk) / CF_SK_SLABS, a.nchunks / CF_SK_SLABS, 256);
          }
        }
      }
      __syncthreads();
      slot = slot1;
    }
    // (the loop's closing barrier retired every LDS read before any wave reuses LDS in its epilogue)
    if constexpr (SK) {
      if (a.nsplit == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = tot[mi][ni];
      }
    }
  }

  if constexpr (SK) {
    if (a.nsplit > 1) {
      f32x4 v[MI * NI * 4];
      if (!cf_splitk_finish(v, a.ws, a.counters, sk_tile, a.nchunks / CF_SK_SLABS, a.nsplit, 256, smem)) return;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][ni][q * 4 + e] = v[(mi * NI + ni) * 4 + q][e];
      __syncthreads();  // the flag word lives in the LDS region the epilogue stages through
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  // The accumulator layout (lane = one output channel, 16 rows per register file) would need 64 scalar stores
  // (+64 residual loads) per lane and the store tail is ISSUE-bound.  Instead each wave transposes its tile through
  // LDS (free after the main loop) 32 rows at a time and touches HBM with 16-byte accesses: lane -> 4 consecutive
  // channels of one pixel; bias / residual / SFT / GELU and the GroupNorm statistics are applied in that layout.
  constexpr int LDW = NI * 32 + 4;       // padded row of the per-wave transpose buffer
  constexpr int Q = NI * 8;              // float4 per tile row
  constexpr int RPP = 64 / Q;            // rows per pass
  constexpr int PASSES = 32 / RPP;
  const bool vec_ok = !(TAPS == 9 && a.out_nchw) && (a.cout % 4) == 0;

  auto epilogue_vec = [&](auto mode) {
    constexpr int EPI = decltype(mode)::value;
    float* stage = smem + wave * (32 * LDW);
    const int cq = lane % Q, rl = lane / Q;
    const int n = n0 + wn * (NI * 32) + cq * 4;
    const bool nvalid = n < a.cout;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && nvalid) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n);
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      size_t offs[PASSES];
      [[maybe_unused]] unsigned inside = 0;  // EXT: bit p = pass p's pixel lies inside the image (edge tiles)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int row = wm * (MI * 32) + mi * 32 + p * RPP + rl;
        size_t pixel;
        if (TAPS == 4)
          pixel = ((size_t)b * a.hout + (2 * (y0 + (row >> 4)) + sub_y)) * a.wout + (2 * (x0 + (row & 15)) + sub_x);
        else if (TAPS == 9)
          pixel = ((size_t)b * a.hout + (y0 + (row >> 4))) * a.wout + (x0 + (row & 15));
        else
          pixel = (size_t)m0 + row;
        offs[p] = pixel * ext_ld<EXT, SK>(a, 2, a.cout) + n;
        if constexpr (EXT) {
          const int oy = TAPS == 4 ? 2 * (y0 + (row >> 4)) + sub_y : y0 + (row >> 4);
          const int ox = TAPS == 4 ? 2 * (x0 + (row & 15)) + sub_x : x0 + (row & 15);
          if (oy < a.hout && ox < a.wout) inside |= 1u << p;
        }
      }
#define CF_LIVE(p) (nvalid && (!EXT || ((inside >> (p)) & 1u)))
#if CF_EPI_WAVESYNC
      // Residual / SFT operands are requested BEFORE the transpose so their HBM latency overlaps it.  The transpose
      // buffer is private to the wave (LDS operations of one wave complete in issue order), so no workgroup barrier is
      // needed: the main loop's closing barrier already retired every read of this LDS region.
      f32x4 r0[PASSES], r1[PASSES];
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        r0[p] = r1[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BIO) {
          if (CF_LIVE(p) && (EPI == CF_EPI_RESIDUAL || EPI == CF_EPI_SFT)) r0[p] = cf_load4_bf16(a.res, offs[p]);
          if (CF_LIVE(p) && EPI == CF_EPI_SFT) r1[p] = cf_load4_bf16(a.sft_scale, offs[p]);
        } else {
        if (CF_LIVE(p) && (EPI == CF_EPI_RESIDUAL || EPI == CF_EPI_SFT || EPI == CF_EPI_AXPY || EPI == CF_EPI_AXPY2))
          r0[p] = *reinterpret_cast<const f32x4*>(a.res + offs[p]);
        if (CF_LIVE(p) && (EPI == CF_EPI_SFT || EPI == CF_EPI_AXPY2)) r1[p] = *reinterpret_cast<const f32x4*>(a.sft_scale + offs[p]);
        }
      }
      __builtin_amdgcn_wave_barrier();
#else
      __syncthreads();  // main loop (or the previous half) is done with this LDS region
#endif
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[cf_acc_row(r, lane) * LDW + ni * 32 + l31] = acc[mi][ni][r];
#if CF_EPI_WAVESYNC
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
      __syncthreads();
#endif
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int trow = p * RPP + rl;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + trow * LDW + cq * 4);
        const size_t o = offs[p];
        if (CF_LIVE(p)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bias4[e];
          if (EPI == CF_EPI_RESIDUAL) {
#if CF_EPI_WAVESYNC
            const f32x4 rr = r0[p];
#else
            const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + o);
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
          } else if (EPI == CF_EPI_SFT) {
#if CF_EPI_WAVESYNC
            const f32x4 dec = r0[p], sc = r1[p];
#else
            const f32x4 dec = *reinterpret_cast<const f32x4*>(a.res + o);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.sft_scale + o);
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = dec[e] + a.sft_w * (dec[e] * sc[e] + v[e]);
          } else if (EPI == CF_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
          } else if (EPI == CF_EPI_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
          } else if (EPI == CF_EPI_AXPY || EPI == CF_EPI_AXPY2) {
            // separately rounded multiply and add, the two ATen ops of `x5 * 0.2 + x` (rrdbnet_arch.py:39,62) -- no FMA contraction
#if CF_EPI_WAVESYNC
            const f32x4 x = r0[p], xx = r1[p];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = __fadd_rn(__fmul_rn(v[e], a.sft_w), x[e]);
              if (EPI == CF_EPI_AXPY2) v[e] = __fadd_rn(__fmul_rn(v[e], a.sft_w), xx[e]);
            }
#endif
          }
          if constexpr (BIO) cf_store4_bf16(a.out, o, v);   // (rounded here, once; the statistics below see the fp32 values)
          else *reinterpret_cast<f32x4*>(a.out + o) = v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ssum[e] += v[e];
            ssq[e] += v[e] * v[e];
          }
        }
      }
    }
    if (!EXT && a.stats_out) {
      // GroupNorm statistics of the values just written, for the NEXT norm: one fp64 partial per (image, group,
      // tile, wave row), combined in a fixed shuffle order -> the later finalize is a deterministic sum.
      const int cpg = a.stats_cpg;
      double d0, q0, d1 = 0, q1 = 0;
      if (cpg == 2) {  // two groups per lane
        d0 = (double)ssum[0] + ssum[1];
        q0 = (double)ssq[0] + ssq[1];
        d1 = (double)ssum[2] + ssum[3];
        q1 = (double)ssq[2] + ssq[3];
      } else {
        d0 = ((double)ssum[0] + ssum[1]) + ((double)ssum[2] + ssum[3]);
        q0 = ((double)ssq[0] + ssq[1]) + ((double)ssq[2] + ssq[3]);
      }
      for (int o = Q; o < 64; o <<= 1) {  // lanes holding the same channels, different rows
        d0 += __shfl_xor(d0, o, 64);
        q0 += __shfl_xor(q0, o, 64);
        d1 += __shfl_xor(d1, o, 64);
        q1 += __shfl_xor(q1, o, 64);
      }
      for (int o = 1; o * 4 < cpg; o <<= 1) {  // adjacent channel quads of one group (cpg >= 8)
        d0 += __shfl_xor(d0, o, 64);
        q0 += __shfl_xor(q0, o, 64);
      }
      if (rl == 0 && nvalid && (n % cpg) == 0) {
        const int tile_in_img = (TAPS > 1) ? (mt - b * a.tiles_per_img) : (m0 - b * (a.hout * a.wout)) / C::BM;
        const size_t pidx = (size_t)tile_in_img * WM + wm;
        const int ng = a.cout / cpg;
        double* o = a.stats_out + (((size_t)b * ng + n / cpg) * a.nparts + pidx) * 2;
        o[0] = d0;
        o[1] = q0;
        if (cpg == 2) {
          o[(size_t)a.nparts * 2] = d1;      // group n/2 + 1
          o[(size_t)a.nparts * 2 + 1] = q1;
        }
      }
    }
  };

#undef CF_LIVE
  // scalar path: NCHW scatter of the 3-channel image (and any cout that is not a multiple of 4)
  auto epilogue_scalar = [&](auto mode) {
    constexpr int EPI = decltype(mode)::value;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
      const bool nvalid = n < a.cout;
      const float bias = (a.bias && nvalid) ? a.bias[n] : 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * (MI * 32) + mi * 32 + cf_acc_row(r, lane);
          size_t pixel;
          int oy = 0, ox = 0;
          if (TAPS == 9) {
            oy = y0 + (row >> 4);
            ox = x0 + (row & 15);
            pixel = ((size_t)b * a.hout + oy) * a.wout + ox;
          } else {
            pixel = (size_t)m0 + row;
          }
          const size_t o = pixel * a.cout + n;
          float v = acc[mi][ni][r] + bias;
          if (nvalid) {
            if (EPI == CF_EPI_RESIDUAL) {
              v += a.res[o];
            } else if (EPI == CF_EPI_SFT) {
              const float dec = a.res[o];
              v = dec + a.sft_w * (dec * a.sft_scale[o] + v);
            } else if (EPI == CF_EPI_GELU) {
              v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            }
            if (TAPS == 9 && a.out_nchw)
              a.out[(((size_t)b * a.cout + n) * a.hout + oy) * a.wout + ox] = v;
            else
              a.out[o] = v;
          }
        }
      }
    }
  };
  if constexpr (EXT) {  // host guarantees NHWC output with cout % 4 == 0
    switch (a.epilogue) {
      case CF_EPI_RESIDUAL: epilogue_vec(std::integral_constant<int, CF_EPI_RESIDUAL>{}); break;
      case CF_EPI_LEAKY: epilogue_vec(std::integral_constant<int, CF_EPI_LEAKY>{}); break;
      case CF_EPI_AXPY: epilogue_vec(std::integral_constant<int, CF_EPI_AXPY>{}); break;
      case CF_EPI_AXPY2: epilogue_vec(std::integral_constant<int, CF_EPI_AXPY2>{}); break;
      default: epilogue_vec(std::integral_constant<int, CF_EPI_NONE>{}); break;
    }
  } else if (vec_ok) {
    switch (a.epilogue) {
      case CF_EPI_RESIDUAL: epilogue_vec(std::integral_constant<int, CF_EPI_RESIDUAL>{}); break;
      case CF_EPI_SFT: epilogue_vec(std::integral_constant<int, CF_EPI_SFT>{}); break;
      case CF_EPI_GELU: epilogue_vec(std::integral_constant<int, CF_EPI_GELU>{}); break;
      default: epilogue_vec(std::integral_constant<int, CF_EPI_NONE>{}); break;
    }
  } else {
    epilogue_scalar(std::integral_constant<int, CF_EPI_NONE>{});  // host guarantees: no epilogue op, no statistics
  }
}

// ---- 3x3 convolution with <= 4 output channels written NCHW (the network's last conv, 64 -> 3 @ 512x512) -----------
// A 3-wide N would waste 29/32 of every MFMA tile (the MFMA instantiation spends 1.24 ms per 16 faces on padding), and the
// layer is HBM-bound anyway (reads 64 channels, writes 3): one thread per output pixel on the vector ALU, the same LDS
// halo patch + prologue as the MFMA kernel, weights fetched through the scalar cache (wave-uniform addresses), coalesced
// per-plane NCHW stores.  Accumulation order: slab, tap, channel (a plain fp32 FMA chain).
// NCO = output channels evaluated per pixel (3 for the RGB head: the zero padding row of the packed weight is not multiplied through)
template <int NCO, bool BIO = false>   // BIO: the input is a bf16 tensor (cf_conv_desc.io_bf16; the NCHW image stays fp32)
__global__ __launch_bounds__(256) void conv3x3_few_cout_kernel(const ConvArgsExt a) {
  constexpr int TH = 16, TW = 16, HWD = TW + 2, NPIX = (TH + 2) * HWD, APT = (NPIX * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[NPIX * CF_LDK];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.tiles_per_img;
  const int r = blockIdx.x - b * a.tiles_per_img;
  const int ty = r / a.tiles_x;
  const int y0 = ty * TH, x0 = (r - ty * a.tiles_x) * TW;
  const int k4 = tid & 3;
  int pix[APT];
#pragma unroll
  for (int j = 0; j < APT; ++j) {
    const int p = (tid >> 2) + 64 * j;
    int v = -1;
    if (p < NPIX) {
      const int hy = p / HWD, hx = p - hy * HWD;
      int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      if (a.pad_mode != CF_PAD_ZERO) {
        iy = cf_border(iy, a.hin, a.pad_mode);
        ix = cf_border(ix, a.win, a.pad_mode);
      }
      if (iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) v = (b * a.hin + iy) * a.win + ix;
    }
    pix[j] = v;
  }
  const int py = tid >> 4, px = tid & 15;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool affine = a.prologue == CF_PRO_AFFINE || a.prologue == CF_PRO_AFFINE_SWISH;
  // Two 16-channel slabs per gather (round 5).  The slab of a pixel is 64 bytes -- half a 128-byte line -- and with one slab per pass the
  // other half was requested a whole tap loop later, when the line had left the L2 (counters: 2.2x the tensor's bytes from HBM); and the
  // loads sat under the per-item validity branch, so each was waited for before the next was issued.  Now both halves of a line are
  // requested together, unconditionally from clamped addresses (out-of-image items are zeroed at the store), the second slab waits in
  // registers while the first is multiplied: the LDS footprint -- what sets the six resident workgroups per CU -- is unchanged, and so is
  // the accumulation order (slab, tap, channel): bitwise the same output.
  for (int chunk = 0; chunk < a.nchunks; chunk += 2) {
    const bool two = chunk + 1 < a.nchunks;
    f32x4 raw[2][APT], sc[2], sh[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = (two || u == 0 ? chunk + u : chunk) * CF_BK + k4 * 4;
      sc[u] = f32x4{1.f, 1.f, 1.f, 1.f};
      sh[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (affine) {
        sc[u] = *reinterpret_cast<const f32x4*>(a.pro_scale + (size_t)b * a.cin + c);
        sh[u] = *reinterpret_cast<const f32x4*>(a.pro_shift + (size_t)b * a.cin + c);
      }
#pragma unroll
      for (int j = 0; j < APT; ++j) {
        const size_t pj = pix[j] < 0 ? (size_t)b * a.hin * a.win : (size_t)pix[j];   // (clamped: any pixel of this image)
        if constexpr (BIO) raw[u][j] = cf_load4_bf16(a.in0, pj * a.c0 + c);
        else raw[u][j] = *reinterpret_cast<const f32x4*>(a.in0 + pj * a.c0 + c);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
#pragma unroll
      for (int j = 0; j < APT; ++j) {
        const int p = (tid >> 2) + 64 * j;
        if (p < NPIX) {
          f32x4 v = raw[u][j];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y = v[e] * sc[u][e] + sh[u][e];
            if (a.prologue == CF_PRO_AFFINE_SWISH) y = swishf(y);
            if (a.prologue == CF_PRO_LEAKY) y = v[e] > 0.f ? v[e] : 0.2f * v[e];
            v[e] = pix[j] >= 0 ? y : 0.f;
          }
          *reinterpret_cast<f32x4*>(As + p * CF_LDK + k4 * 4) = v;
        }
      }
      __syncthreads();
#pragma unroll 1  // one tap's 64 weights fit the scalar registers; unrolling all nine spills them
      for (int tap = 0; tap < 9; ++tap) {
        const float* ap = As + ((py + tap / 3) * HWD + px + tap % 3) * CF_LDK;
        const float* wp = a.weight + (size_t)(tap * a.nchunks + chunk + u) * a.cout_pad * CF_BK;  // [cout_pad][16], wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(ap + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int co = 0; co < NCO; ++co)  // (NCO = 4: rows >= cout of the packed weight are zero padding, no branch needed)
              acc[co] = fmaf(av[e], wp[co * CF_BK + q * 4 + e], acc[co]);
        }
      }
      __syncthreads();
    }
  }
  const int oy = y0 + py, ox = x0 + px;
#pragma unroll
  for (int co = 0; co < 4; ++co)
    if (co < a.cout && oy < a.hout && ox < a.wout)
      a.out[(((size_t)b * a.cout + co) * a.hout + oy) * a.wout + ox] = acc[co] + (a.bias ? a.bias[co] : 0.f);
}

// ---- 3x3 convolution of an NCHW input with <= 4 channels into 64 NHWC channels (the network's first conv, vqgan_arch.py:243) --------
// K = 27: on MFMA the layer pads K to 16 per tap and still moves 1.07 GB of output per 16 faces -- it is write-bound (AI = 13 FLOP/B),
// and the MFMA instantiation reached 1.0 TB/s.  Here a workgroup owns a 16x16 pixel tile: input patch (<= 4 x 18 x 18) and the 27 x 64
// weights sit in LDS, a thread computes FOUR channels of one pixel on the vector ALU (exact fp32 FMA chain: channel-major, then
// taps), and the 16 lanes that share a pixel store its 256 contiguous bytes.  GroupNorm partials as in the MFMA epilogue
// (fp64, fixed shuffle order): one per (image, group, tile, wave).
// C0 = the input channel count when it is known at compile time (3: the network input), 0 = any count <= 4 at run time.  With C0 known
// the thread keeps the 27 float4 weight rows of ITS channel quad in registers for all sixteen pixels it computes: the LDS-resident
// weights cost one ds_read_b128 per tap and pixel, 3456 LDS wave-instructions per tile = 0.31 of the kernel's 0.39 ms per 16 faces.
// The FMA chain (channel-major, then taps) and with it every bit of the output is unchanged.
template <int C0, bool NT = false>   // NT: non-temporal output stores (cf_common.h: cf_store16; a compile-time choice here -- sixteen stores per thread)
__global__ __launch_bounds__(256) void conv3x3_few_cin_kernel(const ConvArgsExt a) {
  __shared__ float s_in[4][18 * 18];
  __shared__ __attribute__((aligned(16))) float s_w[36][64];  // [tap * 4 + c][n]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.tiles_per_img;
  const int r = blockIdx.x - b * a.tiles_per_img;
  const int ty0 = r / a.tiles_x;
  const int y0 = ty0 * 16, x0 = (r - ty0 * a.tiles_x) * 16;
  const int quad = tid & 15, slot = tid >> 4;
  const int n = quad * 4;
  for (int i = tid; i < 36 * 64; i += 256) {
    const int nn = i & 63, tc = i >> 6, tap = tc >> 2, c = tc & 3;
    (&s_w[0][0])[i] = (c < a.c0 && nn < a.cout) ? a.weight[((size_t)tap * a.cout_pad + nn) * CF_BK + c] : 0.f;  // [tap][1 slab][cout_pad][16]
  }
  const size_t plane = (size_t)a.hin * a.win;
  for (int i = tid; i < 4 * 18 * 18; i += 256) {
    const int c = i / (18 * 18), p = i - c * (18 * 18), hy = p / 18, hx = p - hy * 18;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    float v = 0.f;
    if (c < a.c0 && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) v = a.in0[((size_t)b * a.c0 + c) * plane + (size_t)iy * a.win + ix];
    s_in[c][p] = v;
  }
  __syncthreads();
  f32x4 wreg[C0 > 0 ? C0 * 9 : 1];
  if constexpr (C0 > 0) {  // this thread's 27 weight rows: read from LDS once, kept for its sixteen pixels
#pragma unroll
    for (int c = 0; c < C0; ++c)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wreg[c * 9 + tap] = *reinterpret_cast<const f32x4*>(&s_w[tap * 4 + c][n]);
  }
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n);
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int round = 0; round < 16; ++round) {
    const int p = round * 16 + slot, py = p >> 4, px = p & 15;
    f32x4 acc = bias4;
    if constexpr (C0 > 0) {
#pragma unroll
      for (int c = 0; c < C0; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float v = s_in[c][(py + tap / 3) * 18 + px + tap % 3];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, wreg[c * 9 + tap][e], acc[e]);
        }
      }
    } else {
      for (int c = 0; c < a.c0; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float v = s_in[c][(py + tap / 3) * 18 + px + tap % 3];
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(&s_w[tap * 4 + c][n]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, w4[e], acc[e]);
        }
      }
    }
    cf_store16(a.out + (((size_t)b * a.hout + (y0 + py)) * a.wout + (x0 + px)) * a.cout + n, acc, NT);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ssum[e] += acc[e];
      ssq[e] += acc[e] * acc[e];
    }
  }
  if (a.stats_out) {  // cout = 64 -> 2 channels per group: a lane's four channels are two groups
    double d0 = (double)ssum[0] + ssum[1], q0 = (double)ssq[0] + ssq[1];
    double d1 = (double)ssum[2] + ssum[3], q1 = (double)ssq[2] + ssq[3];
    for (int o = 16; o < 64; o <<= 1) {  // the four pixel slots of the wave that share this channel quad
      d0 += __shfl_xor(d0, o, 64);
      q0 += __shfl_xor(q0, o, 64);
      d1 += __shfl_xor(d1, o, 64);
      q1 += __shfl_xor(q1, o, 64);
    }
    if ((tid & 63) < 16) {
      const size_t pidx = (size_t)r * 4 + (tid >> 6);
      double* o = a.stats_out + (((size_t)b * 32 + quad * 2) * a.nparts + pidx) * 2;
      o[0] = d0;
      o[1] = q0;
      o[(size_t)a.nparts * 2] = d1;
      o[(size_t)a.nparts * 2 + 1] = q1;
    }
  }
}

template <int TAPS, int STRIDE, int WM, int WN, int MI, int NI, bool IN_NCHW, bool BF16 = false, bool EXT = false, bool F16 = false, bool BIO = false>
int launch(const ConvArgsExt& a, hipStream_t stream, int* parts_query) {
  using C = Cfg<TAPS, STRIDE, WM, WN, MI, NI>;
  ArgsOf<EXT> k = a;  // (slices the stride fields off for the CodeFormer instantiations)
  int mtiles;
  if (EXT) {  // any image size: ceil-divided tile grid (on the SOURCE grid for the folded upsample), edge tiles masked
    const int gh = TAPS == 4 ? a.hin : a.hout, gw = TAPS == 4 ? a.win : a.wout;
    k.tiles_x = (gw + C::TW - 1) / C::TW;
    k.tiles_per_img = (TAPS == 4 ? 4 : 1) * k.tiles_x * ((gh + C::TH - 1) / C::TH);
    mtiles = k.tiles_per_img * a.batch;
  } else if (TAPS == 4) {  // tiles live on the SOURCE grid; each is computed once per output parity class
    if (a.hin % C::TH != 0 || a.win % C::TW != 0) {
      cf_set_error("cf_conv2d: %dx%d upsample source not divisible by the %dx%d tile", a.hin, a.win, C::TH, C::TW);
      return CF_ERR_ARG;
    }
    k.tiles_x = a.win / C::TW;
    k.tiles_per_img = 4 * k.tiles_x * (a.hin / C::TH);
    mtiles = k.tiles_per_img * a.batch;
  } else if (TAPS == 9) {
    if (a.hout % C::TH != 0 || a.wout % C::TW != 0) {
      cf_set_error("cf_conv2d: %dx%d output not divisible by the %dx%d tile", a.hout, a.wout, C::TH, C::TW);
      return CF_ERR_ARG;
    }
    k.tiles_x = a.wout / C::TW;
    k.tiles_per_img = k.tiles_x * (a.hout / C::TH);
    mtiles = k.tiles_per_img * a.batch;
  } else {
    const long m = (long)a.batch * a.hout * a.wout;
    if (m % C::BM != 0 || (a.hout * a.wout) % C::BM != 0) {
      cf_set_error("cf_conv2d: 1x1 rows %ld (per image %d) not divisible by %d", m, a.hout * a.wout, C::BM);
      return CF_ERR_ARG;
    }
    k.tiles_x = 0;
    k.tiles_per_img = 0;
    mtiles = (int)(m / C::BM);
  }
  k.nparts = (mtiles / a.batch) * WM;  // statistics partials per (image, group): tiles per image x wave rows
  if (parts_query) {
    *parts_query = k.nparts;
    return CF_OK;
  }
  k.ntn = a.cout_pad / C::BN;
  constexpr auto kern = igemm_kernel<TAPS, STRIDE, WM, WN, MI, NI, IN_NCHW, BF16, EXT, F16, false, BIO>;
  constexpr size_t lds = C::LDS_FLOATS * sizeof(float);
  CF_LDS_ATTR(kern, lds);  // (cf_device_init sets the dynamic-LDS attribute on each device)
  hipLaunchKernelGGL(kern, dim3(mtiles * k.ntn), dim3(256), lds, stream, k);
  CF_CHECK_LAUNCH("cf_conv2d");
  return CF_OK;
}

// 1x1 / Linear with split-K (small token matrices): 64x64 tiles, `nsplit` workgroups per tile.
template <int WM, int WN, int MI, int NI>
int launch_sk(const ConvArgsExt& a, float* ws, unsigned* counters, int nsplit, hipStream_t stream, int* parts_query) {
  using C = Cfg<1, 1, WM, WN, MI, NI>;
  ConvArgsSK k;
  static_cast<ConvArgs&>(k) = a;
  const long m = (long)a.batch * a.hout * a.wout;
  if (m % C::BM != 0 || (a.hout * a.wout) % C::BM != 0) {
    cf_set_error("cf_conv2d: 1x1 rows %ld (per image %d) not divisible by %d", m, a.hout * a.wout, C::BM);
    return CF_ERR_ARG;
  }
  const int mtiles = (int)(m / C::BM);
  k.tiles_x = k.tiles_per_img = 0;
  k.nparts = (mtiles / a.batch) * WM;
  if (parts_query) {
    *parts_query = k.nparts;
    return CF_OK;
  }
  k.ntn = a.cout_pad / C::BN;
  k.ws = ws;
  k.counters = counters;
  k.nsplit = nsplit;
  constexpr auto kern = igemm_kernel<1, 1, WM, WN, MI, NI, false, false, false, false, true>;
  constexpr size_t lds = C::LDS_FLOATS * sizeof(float);
  CF_LDS_ATTR(kern, lds);
  hipLaunchKernelGGL(kern, dim3(mtiles * k.ntn * nsplit), dim3(256), lds, stream, k);
  CF_CHECK_LAUNCH("cf_conv2d(split-K)");
  return CF_OK;
}

// Value of packed slab entry (slab, n, c).  slab < 9 (or 1): the plain tap.  Folded nearest-x2 + 3x3 (fold != 0): slab =
// class*4 + tap2 with class = (oy&1)*2 + (ox&1), tap2 = ty*2 + tx over the 2x2 source footprint; the entry is the SUM of the
// 3x3 taps that land on that source pixel: parity 0 -> {k=0} , {k=1,2} ; parity 1 -> {k=0,1} , {k=2} (per axis), summed
// ky-major in fp32 (one rounding per add, fixed order).
__device__ __forceinline__ float packed_weight_value(const float* __restrict__ w, int cout, int cin, int taps, int fold,
                                                     int slab, int n, int c) {
  if (n >= cout || c >= cin) return 0.f;
  const float* wk = w + ((long)n * cin + c) * taps;
  if (!fold) return wk[slab];
  const int cls = slab >> 2, t2 = slab & 3;
  const int sy = cls >> 1, sx = cls & 1, ty = t2 >> 1, tx = t2 & 1;
  const int ky0 = sy == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = sy == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
  const int kx0 = sx == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = sx == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
  float v = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) v += wk[ky * 3 + kx];
  return v;
}

__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int cin, int taps, int fold, int cout_pad,
                                   int nchunks, float* __restrict__ packed, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % CF_BK);
  long r = i / CF_BK;
  const int n = (int)(r % cout_pad);
  r /= cout_pad;
  const int chunk = (int)(r % nchunks);
  const int slab = (int)(r / nchunks);
  packed[i] = packed_weight_value(w, cout, cin, taps, fold, slab, n, chunk * CF_BK + k);
}

// 16-bit variants: [tap][cin_pad/32][cout_pad][32] bf16 or IEEE half (round-to-nearest-even), two values per 32-bit word.
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, int cout, int cin, int taps, int fold, int cout_pad,
                                        int nchunks, unsigned* __restrict__ packed, long total_words, int f16) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total_words) return;
  const int k2 = (int)(i % 16);  // word index inside the 32-channel row
  long r = i / 16;
  const int n = (int)(r % cout_pad);
  r /= cout_pad;
  const int chunk = (int)(r % nchunks);
  const int slab = (int)(r / nchunks);
  unsigned out = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float v = packed_weight_value(w, cout, cin, taps, fold, slab, n, chunk * 32 + k2 * 2 + h);
    if (f16) {
      const _Float16 hv = (_Float16)v;
      out |= (unsigned)__builtin_bit_cast(unsigned short, hv) << (16 * h);
      continue;
    }
    unsigned u = __builtin_bit_cast(unsigned, v);
    u += 0x7fffu + ((u >> 16) & 1u);
    out |= (u >> 16) << (16 * h);
  }
  packed[i] = out;
}

}  // namespace

static int pack_bf16(const float* w, int cout, int cin, int fold, int cout_pad, int cin_pad, void* packed, cf_stream_t stream,
                     int f16 = 0) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_bf16/f16: null pointer");
  CF_REQUIRE(cin_pad % 32 == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % (f16 ? 32 : 64) == 0,
             "cf_pack_conv_weight_bf16/f16: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  const int slabs = fold ? 16 : 9;
  const long words = (long)slabs * cin_pad * cout_pad / 2;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout,
                     cin, 9, fold, cout_pad, cin_pad / 32, reinterpret_cast<unsigned*>(packed), words, f16);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_bf16");
  return CF_OK;
}

extern "C" int cf_pack_conv_weight_bf16(const float* w, int cout, int cin, int taps, int cout_pad, int cin_pad, void* packed,
                                        cf_stream_t stream) {
  CF_REQUIRE(taps == 9, "cf_pack_conv_weight_bf16: the bf16 path covers 3x3 convolutions (taps=9)");
  return pack_bf16(w, cout, cin, 0, cout_pad, cin_pad, packed, stream);
}

extern "C" int cf_pack_conv_weight_up2x_bf16(const float* w, int cout, int cin, int cout_pad, int cin_pad, void* packed,
                                             cf_stream_t stream) {
  return pack_bf16(w, cout, cin, 1, cout_pad, cin_pad, packed, stream);
}

extern "C" int cf_pack_conv_weight_f16(const float* w, int cout, int cin, int taps, int cout_pad, int cin_pad, void* packed,
                                       cf_stream_t stream) {
  CF_REQUIRE(taps == 9, "cf_pack_conv_weight_f16: 3x3 weights only (taps=%d)", taps);
  return pack_bf16(w, cout, cin, 0, cout_pad, cin_pad, packed, stream, 1);
}

extern "C" int cf_pack_conv_weight_up2x_f16(const float* w, int cout, int cin, int cout_pad, int cin_pad, void* packed,
                                            cf_stream_t stream) {
  return pack_bf16(w, cout, cin, 1, cout_pad, cin_pad, packed, stream, 1);
}

extern "C" int cf_pack_conv_weight_up2x(const float* w, int cout, int cin, int cout_pad, int cin_pad, float* packed,
                                        cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_up2x: null pointer");
  CF_REQUIRE(cin_pad % CF_BK == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 32 == 0,
             "cf_pack_conv_weight_up2x: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  const long total = 16L * cin_pad * cout_pad;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                     9, 1, cout_pad, cin_pad / CF_BK, packed, total);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_up2x");
  return CF_OK;
}

extern "C" int64_t cf_packed_weight_elems(int cin_pad, int taps, int cout_pad) {
  return (int64_t)taps * cin_pad * cout_pad;
}

extern "C" int cf_pack_conv_weight(const float* w, int cout, int cin, int taps, int cout_pad, int cin_pad,
                                   float* packed, cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight: null pointer");
  CF_REQUIRE(taps == 1 || taps == 9, "cf_pack_conv_weight: taps must be 1 or 9 (got %d)", taps);
  CF_REQUIRE(cin_pad % CF_BK == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 32 == 0,
             "cf_pack_conv_weight: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  const long total = (long)taps * cin_pad * cout_pad;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, taps, 0,
                     cout_pad, cin_pad / CF_BK, packed, total);
  CF_CHECK_LAUNCH("cf_pack_conv_weight");
  return CF_OK;
}

int cf_winograd_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query);  // cf_winograd.hip
int cf_wf43_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query);      // cf_wf43.hip
int cf_gemm_split_launch(const cf_conv_desc* d, hipStream_t stream);                     // cf_gemm_split.hip
int cf_gemm_split_geometry(const cf_conv_desc* d, int* tiles, long* bytes_per_part);
int cf_gemm_f32_tile_try(const cf_conv_desc* d, hipStream_t stream);                     // cf_gemm_split.hip: fp32 token tiles (CF_OK: launched, 1: not its shape)
int cf_split_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query);     // cf_split.hip

static int conv_dispatch(const cf_conv_desc* d, hipStream_t stream, int* pq) {
  CF_REQUIRE(d, "cf_conv2d: null descriptor");
  CF_REQUIRE(pq || (d->in0 && d->weight && d->out), "cf_conv2d: null in0/weight/out");
  CF_REQUIRE(d->taps == 1 || d->taps == 9, "cf_conv2d: taps must be 1 or 9 (got %d)", d->taps);
  CF_REQUIRE(d->stride == 1 || (d->stride == 2 && d->taps == 9), "cf_conv2d: unsupported stride %d", d->stride);
  CF_REQUIRE(d->batch > 0 && d->hin > 0 && d->win > 0 && d->cout > 0, "cf_conv2d: bad dims");
  CF_REQUIRE(!(d->upsample && (d->stride != 1 || d->taps != 9 || d->in_nchw || d->out_nchw)),
             "cf_conv2d: upsample needs a 3x3 stride-1 NHWC conv");
  const int exp_h = d->stride == 2 ? d->hin / 2 : (d->hin << (d->upsample ? 1 : 0));
  const int exp_w = d->stride == 2 ? d->win / 2 : (d->win << (d->upsample ? 1 : 0));
  CF_REQUIRE(d->hout == exp_h && d->wout == exp_w, "cf_conv2d: hout/wout %dx%d, expected %dx%d", d->hout, d->wout,
             exp_h, exp_w);
  CF_REQUIRE(d->stride == 1 || (d->hin % 2 == 0 && d->win % 2 == 0), "cf_conv2d: stride 2 needs even input");
  if (d->out_nchw || d->cout % 4 != 0)
    CF_REQUIRE(d->epilogue == CF_EPI_NONE && !d->stats_out, "cf_conv2d: cout %d / NCHW output supports no epilogue op or statistics",
               d->cout);
  if (d->stats_out || (pq && d->stats_cpg)) {
    const int g = d->stats_cpg;
    CF_REQUIRE(g >= 2 && g <= 32 && (g & (g - 1)) == 0 && d->cout % g == 0 && !d->out_nchw,
               "cf_conv2d: stats_cpg %d must be a power of two in [2,32] dividing cout %d (NHWC output)", g, d->cout);
  }
  if (d->in_nchw) {
    CF_REQUIRE(d->c0 >= 1 && d->c0 <= 4 && d->c1 == 0 && d->taps == 9 && d->stride == 1 && !d->upsample,
               "cf_conv2d: in_nchw supports 3x3 s1 with <=4 input channels");
    CF_REQUIRE(d->prologue == CF_PRO_NONE, "cf_conv2d: in_nchw has no prologue");
  } else {
    CF_REQUIRE(d->c0 % CF_BK == 0 && d->c1 % CF_BK == 0 && d->c0 > 0, "cf_conv2d: c0=%d c1=%d must be multiples of 16",
               d->c0, d->c1);
    CF_REQUIRE(d->c1 == 0 || d->in1, "cf_conv2d: c1 > 0 without in1");
  }
  CF_REQUIRE(d->bf16_mfma >= CF_OPERAND_F32 && d->bf16_mfma <= CF_OPERAND_F16X2, "cf_conv2d: bad operand format %d", d->bf16_mfma);
  if (d->bf16_mfma == CF_OPERAND_BF16 || d->bf16_mfma == CF_OPERAND_F16)
    CF_REQUIRE(d->taps == 9 && d->stride == 1 && !d->in_nchw && !d->out_nchw && d->c0 % 32 == 0 && d->c1 % 32 == 0 &&
                   d->cout_pad % (d->bf16_mfma == CF_OPERAND_F16 ? 32 : 64) == 0 && d->cout % 4 == 0,
               "cf_conv2d: bf16_mfma covers 3x3 stride-1 NHWC convs with channels %% 32 == 0 (c0=%d c1=%d cout_pad=%d)", d->c0,
               d->c1, d->cout_pad);
  CF_REQUIRE(d->prologue >= 0 && d->prologue <= 3 && d->epilogue >= 0 && d->epilogue <= CF_EPI_AXPY2, "cf_conv2d: bad pro/epilogue");
  if (d->prologue == CF_PRO_AFFINE || d->prologue == CF_PRO_AFFINE_SWISH)
    CF_REQUIRE(d->pro_scale && d->pro_shift, "cf_conv2d: affine prologue without scale/shift tables");
  if (d->epilogue == CF_EPI_RESIDUAL || d->epilogue == CF_EPI_SFT || d->epilogue == CF_EPI_AXPY || d->epilogue == CF_EPI_AXPY2)
    CF_REQUIRE(d->res, "cf_conv2d: epilogue needs res");
  if (d->epilogue == CF_EPI_SFT) CF_REQUIRE(d->sft_scale, "cf_conv2d: SFT epilogue needs sft_scale");
  if (d->epilogue == CF_EPI_AXPY2) CF_REQUIRE(d->sft_scale, "cf_conv2d: AXPY2 epilogue needs res2 (sft_scale)");
  // general (EXT) instantiations: strided channel slices, dense-block epilogues, image sizes off the tile grid
  const int ld0 = d->ld_in0 > 0 ? d->ld_in0 : d->c0, ld1 = d->ld_in1 > 0 ? d->ld_in1 : d->c1,
            ldo = d->ld_out > 0 ? d->ld_out : d->cout;
  CF_REQUIRE(d->ld_in0 >= 0 && d->ld_in1 >= 0 && d->ld_out >= 0 && ld0 >= d->c0 && ld1 >= d->c1 && ldo >= d->cout,
             "cf_conv2d: channel strides (%d,%d,%d) smaller than the channel counts (%d,%d,%d)", ld0, ld1, ldo, d->c0, d->c1,
             d->cout);
  const bool few_cout = d->taps == 9 && d->stride == 1 && d->out_nchw && d->cout <= 4 && !d->upsample && d->c1 == 0 && !d->in_nchw;
  CF_REQUIRE(d->pad_mode >= CF_PAD_ZERO && d->pad_mode <= CF_PAD_EDGE && (d->pad_lo == 0 || d->pad_lo == 1),
             "cf_conv2d: bad pad_mode %d / pad_lo %d", d->pad_mode, d->pad_lo);
  CF_REQUIRE(d->pad_mode != CF_PAD_REFLECT || (d->taps == 9 && !d->upsample && d->hin >= 2 && d->win >= 2),
             "cf_conv2d: reflect padding is for plain 3x3 convs on images of at least 2x2");
  CF_REQUIRE(d->pad_mode != CF_PAD_EDGE || d->upsample, "cf_conv2d: edge padding belongs to the folded upsample conv");
  CF_REQUIRE(d->pad_lo == 0 || d->stride == 2, "cf_conv2d: pad_lo applies to stride 2");
  const bool ext = !d->winograd && d->bf16_mfma != CF_OPERAND_F16X2 && (ld0 != d->c0 || ld1 != d->c1 || ldo != d->cout || d->epilogue >= CF_EPI_LEAKY ||
                   ((d->pad_mode != CF_PAD_ZERO || d->pad_lo) && !(d->out_nchw && d->cout <= 4)) ||
                   (d->bf16_mfma == CF_OPERAND_F16 && d->cout_pad % 64 != 0) ||
                   (d->taps == 9 && d->stride == 1 && !d->in_nchw && !few_cout && (d->hout % 16 != 0 || d->wout % 16 != 0)));
  if (ext) {
    CF_REQUIRE(d->taps == 9 && (d->stride == 1 || (d->bf16_mfma == CF_OPERAND_F32 && d->cout_pad % 128 == 0)) && !d->in_nchw &&
                   !d->out_nchw && d->bf16_mfma != CF_OPERAND_BF16 && !d->stats_out &&
                   !(pq && d->stats_cpg) && d->cout % 4 == 0 && d->epilogue != CF_EPI_SFT && d->epilogue != CF_EPI_GELU,
               "cf_conv2d: strided slices / leaky|axpy epilogues / off-grid sizes (%dx%d) need a 3x3 stride-1 fp32 / f16-operand NHWC conv with "
               "cout %% 4 == 0, no statistics, epilogue in {none, residual, leaky, axpy, axpy2}", d->hout, d->wout);
    CF_REQUIRE(ld0 % 4 == 0 && ld1 % 4 == 0 && ldo % 4 == 0, "cf_conv2d: channel strides must be multiples of 4 floats");
  } else {
    CF_REQUIRE(!few_cout || ld0 == d->c0, "cf_conv2d: the NCHW <=4-channel output conv reads a dense input");
  }
  CF_REQUIRE(!(d->out_nchw && (d->taps != 9 || d->epilogue != CF_EPI_NONE)), "cf_conv2d: out_nchw needs 3x3, no epilogue");
  CF_REQUIRE(d->cout_pad >= d->cout && d->cout_pad % 32 == 0, "cf_conv2d: cout_pad %d invalid for cout %d", d->cout_pad,
             d->cout);

  CF_REQUIRE(!d->in0_alt || (d->taps == 1 && d->bf16_mfma == CF_OPERAND_F16X2 && (long)d->hout * d->wout <= CF_TOKEN_IMAGE_MAX && !d->io_bf16),
             "cf_conv2d: in0_alt (a second token matrix for the columns >= alt_cout0) belongs to the split-half token GEMM");
  if (d->io_bf16) {   // bf16 tensors (ABI v22): exactly the launches the bf16 mode makes from 64x64 pixels up; anything else is refused, never reinterpreted
    const bool dense = ld0 == d->c0 && ld1 == d->c1 && ldo == d->cout && d->pad_mode == CF_PAD_ZERO && !d->in_nchw && d->stride == 1 && d->split_k < 1;
    const bool wino_bf16 = d->winograd == 1 && d->bf16_mfma == CF_OPERAND_BF16;
    const bool direct_bf16 = !d->winograd && d->bf16_mfma == CF_OPERAND_BF16 && d->taps == 9 && !d->out_nchw &&
                             (d->upsample ? (d->cout_pad % 128 == 0 && (long)d->hout * d->wout > 1024) : d->cout_pad == 64);
    const bool conv1 = !d->winograd && d->bf16_mfma == CF_OPERAND_F32 && d->taps == 1 && (d->cout_pad % 128 == 0 || d->cout_pad == 64) &&
                       (long)d->hout * d->wout > 1024;
    const bool conv1s = !d->winograd && d->bf16_mfma == CF_OPERAND_F16X2 && d->taps == 1 && (long)d->hout * d->wout > CF_TOKEN_IMAGE_MAX;   // streaming 1x1 (cf_split.hip)
    const bool head = few_cout && d->bf16_mfma == CF_OPERAND_F32 && !d->winograd && d->cout == 3;
    CF_REQUIRE(d->io_bf16 == 1 && dense && !ext && (head || d->cout % 4 == 0) && (wino_bf16 || direct_bf16 || conv1 || conv1s || head),
               "cf_conv2d: io_bf16 (bf16 tensors) is built for dense stride-1 launches of the bf16 mode: winograd + CF_OPERAND_BF16, direct CF_OPERAND_BF16 3x3 "
               "(cout_pad 64) / folded upsample (cout_pad %% 128 == 0), fp32 1x1 on images of more than 1024 pixels, the 3-channel NCHW head "
               "(taps %d operand %d winograd %d cout_pad %d upsample %d)", d->taps, d->bf16_mfma, d->winograd, d->cout_pad, d->upsample);
    CF_REQUIRE(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_RESIDUAL || d->epilogue == CF_EPI_SFT, "cf_conv2d: io_bf16 epilogues are none / residual / SFT");
  }
  if (d->act_scale) {
    CF_REQUIRE(d->prologue == CF_PRO_NONE || d->prologue == CF_PRO_LEAKY,
               "cf_conv2d: act_scale is for un-normalised inputs (prologue none / leaky), got prologue %d", d->prologue);
    const bool conv1x1_split = d->taps == 1 && d->bf16_mfma == CF_OPERAND_F16X2 && (long)d->hout * d->wout > CF_TOKEN_IMAGE_MAX;
    CF_REQUIRE((d->taps == 9 && (d->bf16_mfma == CF_OPERAND_F32 || d->winograd || d->bf16_mfma == CF_OPERAND_F16X2)) || conv1x1_split,
               "cf_conv2d: act_scale is applied by the Winograd / split-half convolution kernels only (operand %d, winograd %d, taps %d)",
               d->bf16_mfma, d->winograd, d->taps);
  }
  if (d->taps == 1 && d->bf16_mfma == CF_OPERAND_F16X2) {
    CF_REQUIRE(!pq, "cf_conv2d(1x1, f16x2): no statistics epilogue");
    // images of more than CF_TOKEN_IMAGE_MAX pixels: the streaming 1x1 form of the split-half convolution kernel (weight: form 3 of
    // cf_pack_conv_weight_f16x2); token matrices (the Transformer's 16x16 "images"): the token GEMM (cf_pack_linear_weight_f16x2)
    if ((long)d->hout * d->wout > CF_TOKEN_IMAGE_MAX) return cf_split_launch(d, stream, pq);
    return cf_gemm_split_launch(d, stream);
  }
  if (d->winograd == 2) return cf_wf43_launch(d, stream, pq);  // F(4x4,3x3), split-half operands (cf_wf43.hip)
  CF_REQUIRE(d->winograd == 0 || d->winograd == 1, "cf_conv2d: winograd must be 0, 1 (F(2x2,3x3)) or 2 (F(4x4,3x3)), got %d", d->winograd);
  if (d->winograd) return cf_winograd_launch(d, stream, pq);  // fp32 or split-half operands
  if (d->bf16_mfma == CF_OPERAND_F16X2) return cf_split_launch(d, stream, pq);

  ConvArgsExt a;
  a.in0 = d->in0;
  a.in1 = d->in1;
  a.c0 = d->c0;
  a.c1 = d->c1;
  a.cin = d->c0 + d->c1;
  a.nchunks = d->bf16_mfma ? a.cin / 32 : (a.cin + CF_BK - 1) / CF_BK;
  a.batch = d->batch;
  a.hin = d->hin;
  a.win = d->win;
  a.hout = d->hout;
  a.wout = d->wout;
  a.cout = d->cout;
  a.cout_pad = d->cout_pad;
  a.upsample = d->upsample ? 1 : 0;
  a.out_nchw = d->out_nchw;
  a.prologue = d->prologue;
  a.epilogue = d->epilogue;
  a.pro_scale = d->pro_scale;
  a.pro_shift = d->pro_shift;
  a.weight = d->weight;
  a.bias = d->bias;
  a.res = d->res;
  a.sft_scale = d->sft_scale;
  a.sft_w = d->sft_w;
  a.out = d->out;
  a.stats_out = d->stats_out;
  a.stats_cpg = d->stats_cpg > 0 ? d->stats_cpg : 1;
  a.nparts = 0;
  a.tiles_x = a.tiles_per_img = a.ntn = 0;
  a.ld0 = ld0;
  a.ld1 = ld1;
  a.ldo = ldo;
  a.pad_mode = d->pad_mode;
  a.pad_lo = d->pad_lo;
  a.nt_out = cf_nt_store((long)d->batch * d->hout * d->wout * d->cout * 4);

  const int cp = d->cout_pad;
  // Small-M layers (16x16 / 32x32 latents): at batch 16 a 128x128 tiling yields only 128-256 workgroups for 256 CUs;
  // halving the N tile doubles the workgroup count at the price of gathering the halo patch twice.  The choice depends
  // on the per-image shape ONLY, never on the batch: tiling (and with it the order of the statistics partials) must be
  // the same for a face whether it is restored alone or inside any batch / shard, so results stay bitwise batch-invariant.
  const bool narrow = cp % 128 == 0 && (long)d->hout * d->wout <= 1024;
  if (ext) {
    const bool f16 = d->bf16_mfma == CF_OPERAND_F16;
    if (d->stride == 2) return launch<9, 2, 2, 2, 2, 2, false, false, true>(a, stream, pq);  // fp32, cout_pad % 128 == 0 (checked)
    if (d->upsample) {
      CF_REQUIRE(cp % 64 == 0, "cf_conv2d: general upsample path needs cout_pad %% 64 == 0 (got %d)", cp);
      if (f16) return launch<4, 1, 4, 1, 2, 2, false, false, true, true>(a, stream, pq);
      if (cp % 128 == 0) return launch<4, 1, 2, 2, 2, 2, false, false, true>(a, stream, pq);
      return launch<4, 1, 4, 1, 2, 2, false, false, true>(a, stream, pq);
    }
    if (f16) {
      if (cp % 64 == 0) return launch<9, 1, 4, 1, 2, 2, false, false, true, true>(a, stream, pq);
      return launch<9, 1, 4, 1, 2, 1, false, false, true, true>(a, stream, pq);
    }
    if (cp % 128 == 0) return launch<9, 1, 2, 2, 2, 2, false, false, true>(a, stream, pq);
    if (cp % 64 == 0) return launch<9, 1, 4, 1, 2, 2, false, false, true>(a, stream, pq);
    return launch<9, 1, 4, 1, 2, 1, false, false, true>(a, stream, pq);  // cout_pad % 32 == 0 (checked above)
  }
  if (d->upsample) {  // nearest x2 + 3x3 as four 2x2 sub-pixel convolutions; weight packed by cf_pack_conv_weight_up2x[_bf16]
    if (d->bf16_mfma == CF_OPERAND_F16) {
      if (narrow) return launch<4, 1, 2, 2, 2, 1, false, false, false, true>(a, stream, pq);
      if (cp % 128 == 0) return launch<4, 1, 2, 2, 2, 2, false, false, false, true>(a, stream, pq);
      return launch<4, 1, 4, 1, 2, 2, false, false, false, true>(a, stream, pq);
    }
    if (d->bf16_mfma && d->io_bf16) return launch<4, 1, 2, 2, 2, 2, false, true, false, false, true>(a, stream, pq);   // (cout_pad % 128 == 0, > 1024 pixels: checked above)
    if (d->bf16_mfma) {
      if (narrow) return launch<4, 1, 2, 2, 2, 1, false, true>(a, stream, pq);
      if (cp % 128 == 0) return launch<4, 1, 2, 2, 2, 2, false, true>(a, stream, pq);
      return launch<4, 1, 4, 1, 2, 2, false, true>(a, stream, pq);
    }
    if (narrow) return launch<4, 1, 2, 2, 2, 1, false>(a, stream, pq);
    if (cp % 128 == 0) return launch<4, 1, 2, 2, 2, 2, false>(a, stream, pq);
    if (cp == 64) return launch<4, 1, 4, 1, 2, 2, false>(a, stream, pq);
    cf_set_error("cf_conv2d: upsample path needs cout_pad 64 or a multiple of 128 (got %d)", cp);
    return CF_ERR_ARG;
  }
  if (d->bf16_mfma == CF_OPERAND_F16) {
    if (narrow) return launch<9, 1, 2, 2, 2, 1, false, false, false, true>(a, stream, pq);
    if (cp % 128 == 0) return launch<9, 1, 2, 2, 2, 2, false, false, false, true>(a, stream, pq);
    return launch<9, 1, 4, 1, 2, 2, false, false, false, true>(a, stream, pq);  // cout_pad == 64
  }
  if (d->bf16_mfma && d->io_bf16) return launch<9, 1, 4, 1, 2, 2, false, true, false, false, true>(a, stream, pq);   // (cout_pad == 64: checked above)
  if (d->bf16_mfma) {
    if (narrow) return launch<9, 1, 2, 2, 2, 1, false, true>(a, stream, pq);
    if (cp % 128 == 0) return launch<9, 1, 2, 2, 2, 2, false, true>(a, stream, pq);
    return launch<9, 1, 4, 1, 2, 2, false, true>(a, stream, pq);  // cout_pad == 64
  }
  if (d->taps == 9 && d->stride == 1) {
    if (d->in_nchw && d->cout == 64 && cp == 64 && d->hout % 16 == 0 && d->wout % 16 == 0 && (!d->stats_cpg || d->stats_cpg == 2) &&
        d->epilogue == CF_EPI_NONE) {  // the vector-ALU first conv (write-bound layer)
      a.tiles_x = d->wout / 16;
      a.tiles_per_img = a.tiles_x * (d->hout / 16);
      a.nparts = a.tiles_per_img * 4;
      if (pq) {
        *pq = a.nparts;
        return CF_OK;
      }
      if (d->c0 == 3 && a.nt_out) hipLaunchKernelGGL((conv3x3_few_cin_kernel<3, true>), dim3(a.tiles_per_img * d->batch), dim3(256), 0, stream, a);
      else if (d->c0 == 3) hipLaunchKernelGGL(conv3x3_few_cin_kernel<3>, dim3(a.tiles_per_img * d->batch), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(conv3x3_few_cin_kernel<0>, dim3(a.tiles_per_img * d->batch), dim3(256), 0, stream, a);
      CF_CHECK_LAUNCH("cf_conv2d");
      return CF_OK;
    }
    if (d->in_nchw) {
      CF_REQUIRE(cp == 64, "cf_conv2d: in_nchw path is built for cout_pad 64 (got %d)", cp);
      return launch<9, 1, 4, 1, 2, 2, true>(a, stream, pq);
    }
    if (narrow) return launch<9, 1, 2, 2, 2, 1, false>(a, stream, pq);
    if (cp % 128 == 0) return launch<9, 1, 2, 2, 2, 2, false>(a, stream, pq);
    if (cp == 64) return launch<9, 1, 4, 1, 2, 2, false>(a, stream, pq);
    if (few_cout && !pq) {  // any image size: edge tiles are masked
      a.tiles_x = (d->wout + 15) / 16;
      a.tiles_per_img = a.tiles_x * ((d->hout + 15) / 16);
      if (d->cout == 3 && d->io_bf16) hipLaunchKernelGGL((conv3x3_few_cout_kernel<3, true>), dim3(a.tiles_per_img * d->batch), dim3(256), 0, stream, a);
      else if (d->cout == 3) hipLaunchKernelGGL(conv3x3_few_cout_kernel<3>, dim3(a.tiles_per_img * d->batch), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(conv3x3_few_cout_kernel<4>, dim3(a.tiles_per_img * d->batch), dim3(256), 0, stream, a);
      CF_CHECK_LAUNCH("cf_conv2d");
      return CF_OK;
    }
    if (cp == 32) return launch<9, 1, 4, 1, 2, 1, false>(a, stream, pq);
  } else if (d->taps == 9 && d->stride == 2) {
    // 128-wide channel tiles unless the layer is small (at most 64 of them per image): 64-wide tiles double the workgroup count of a
    // small batch.  Per-image shape only: the two widths group the statistics partials differently (see cf_split_launch)
    const long wide_wgs = ((long)d->hout * d->wout / 128) * (cp / 128);
    if (cp % 128 == 0 && wide_wgs > 64) return launch<9, 2, 2, 2, 2, 2, false>(a, stream, pq);
    if (cp % 64 == 0) return launch<9, 2, 2, 2, 2, 1, false>(a, stream, pq);
  } else {
    if (d->split_k >= 1) {  // small token matrices: 64x64 tiles, split_k workgroups per tile
      const int V = a.nchunks / CF_SK_SLABS;  // virtual chunks: the summation order is out = ((0 + P0) + P1) + ... whatever split_k is
      CF_REQUIRE(cp % 64 == 0 && a.nchunks % CF_SK_SLABS == 0 && V >= 1 && V % d->split_k == 0,
                 "cf_conv2d: split_k %d needs cout_pad %% 64 == 0, K %% 128 == 0 and split_k dividing K/128 = %d", d->split_k, V);
      CF_REQUIRE(pq || d->split_k == 1 || (d->workspace && d->counters), "cf_conv2d: split_k > 1 needs workspace and counters");
      if (!pq && d->split_k == 1) {   // one workgroup per tile on a large token matrix: the 128-token tile kernel (bitwise the same result)
        const int rc = cf_gemm_f32_tile_try(d, stream);
        if (rc != 1) return rc;
      }
      return launch_sk<2, 2, 1, 1>(a, d->workspace, d->counters, d->split_k, stream, pq);
    }
    if (d->io_bf16) {   // (images of more than 1024 pixels, cout_pad 64 or a multiple of 128: checked above)
      if (cp % 128 == 0) return launch<1, 1, 2, 2, 2, 2, false, false, false, false, true>(a, stream, pq);
      return launch<1, 1, 4, 1, 2, 2, false, false, false, false, true>(a, stream, pq);
    }
    if (narrow) return launch<1, 1, 2, 2, 2, 1, false>(a, stream, pq);
    if (cp % 128 == 0) return launch<1, 1, 2, 2, 2, 2, false>(a, stream, pq);
    if (cp == 64) return launch<1, 1, 4, 1, 2, 2, false>(a, stream, pq);
  }
  cf_set_error("cf_conv2d: no kernel for taps=%d stride=%d cout_pad=%d", d->taps, d->stride, cp);
  return CF_ERR_ARG;
}

extern "C" int cf_conv2d(const cf_conv_desc* d, cf_stream_t stream) { return conv_dispatch(d, (hipStream_t)stream, nullptr); }

// split-K geometry: output tiles of the launch and accumulator bytes one (tile, split) parks in the workspace
int cf_winograd_splitk_geometry(const cf_conv_desc* d, int* tiles, long* bytes_per_part);  // cf_winograd.hip
int cf_split_splitk_geometry(const cf_conv_desc* d, int* tiles, long* bytes_per_part);     // cf_split.hip
static int splitk_geometry(const cf_conv_desc* d, int* tiles, long* bytes_per_part) {
  *tiles = 0;
  *bytes_per_part = 0;
  if (!d || d->split_k < 1) return CF_OK;
  if (d->taps == 1 && d->bf16_mfma == CF_OPERAND_F16X2) return cf_gemm_split_geometry(d, tiles, bytes_per_part);
  if (d->winograd) return cf_winograd_splitk_geometry(d, tiles, bytes_per_part);
  if (d->bf16_mfma == CF_OPERAND_F16X2) return cf_split_splitk_geometry(d, tiles, bytes_per_part);
  CF_REQUIRE(d->taps == 1 && d->cout_pad % 64 == 0 && ((long)d->batch * d->hout * d->wout) % 64 == 0,
             "cf_conv2d: split_k covers 1x1 / Linear (64x64 tiles), winograd and f16x2 launches");
  *tiles = (int)((long)d->batch * d->hout * d->wout / 64) * (d->cout_pad / 64);
  *bytes_per_part = 64L * 64 * 4 * ((d->c0 + d->c1) / (16 * CF_SK_SLABS)) / (d->split_k > 0 ? d->split_k : 1);  // V chunk sums per tile in all
  return CF_OK;
}
extern "C" int64_t cf_conv2d_workspace_bytes(const cf_conv_desc* d) {
  int tiles = 0;
  long per = 0;
  const int rc = splitk_geometry(d, &tiles, &per);
  if (rc != CF_OK) return rc;
  return d && d->split_k > 1 ? (int64_t)tiles * d->split_k * per : 0;
}
extern "C" int cf_conv2d_tiles(const cf_conv_desc* d) {
  int tiles = 0;
  long per = 0;
  const int rc = splitk_geometry(d, &tiles, &per);
  return rc != CF_OK ? rc : tiles;
}

extern "C" int cf_conv2d_stats_parts(const cf_conv_desc* d) {
  int parts = 0;
  const int rc = conv_dispatch(d, nullptr, &parts);
  return rc == CF_OK ? parts : rc;
}
