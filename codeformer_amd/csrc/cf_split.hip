// 3x3 convolution with fp32-grade accuracy on the f16 MFMA pipe (v_mfma_f32_32x32x16_f16) for gfx950: split operands.
//
// The fp32 MFMA runs at the fp32 VECTOR rate (157 TFLOP/s, 1/16 of the 16-bit matrix rate) and shares the FMA pipe with the
// VALU work of the gather, so the exact-fp32 kernels (cf_igemm.hip / cf_winograd.hip) top out near 140 TFLOP/s of issued
// multiplies.  Here every fp32 operand x is written as x = hi + lo with hi = f16(x) and lo = f16(x - hi) (both round-to-nearest-
// even; |x - hi - lo| <= 2^-22 |x|), and a product is evaluated as
//      a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (the dropped a_lo*b_lo term is <= 2^-22 |a b|)
// i.e. THREE f16 MFMAs with fp32 accumulation per fp32 multiply-add: 3/16 of the fp32-MFMA issue time, on a pipe that
// co-executes with the VALU.  Operands carry 22 instead of 24 significant bits; accumulation is fp32.  Measured against fp64
// the error of a layer is 1-3x that of the exact-fp32 kernels (tests/test_gpu_split.py), against the reference the whole
// network stays inside the 1e-3 pixel tolerance by more than an order of magnitude.  What runs on this kernel in the default mode
// (precision 'f16x2'): the folded upsample convolutions, the stride-2 Downsample convolutions and the image-sized 1x1 skip
// convolutions of ENCODER, generator and fusion blocks (the 3x3 stride-1 layers take the Winograd forms of the same split scheme,
// cf_wsplit.hip / cf_winograd.hip / cf_wf43.hip; reference: vqgan_arch.py:117-164,243-323, codeformer_arch.py:136-157).  The encoder
// is therefore NOT on exact fp32 in that mode: logits and code indices are not bitwise those of precision 'fp32' -- they meet the
// reference's goldens (logits 1e-4, indices exact; tests/test_gpu_parity.py, tests/test_gpu_range.py).  Exact fp32 in every mode:
// attention, AttnBlock 1x1, feat_emb, statistics and the code argmax; encoder_precision = 'fp32' together with gemm_precision =
// 'fp32' puts encoder and Transformer back on exact fp32 MFMA.
//
// Range: hi is an IEEE half, so conv INPUTS (post-activation values) must stay below 65504 in magnitude -- an overflow becomes
// inf / NaN in the output, never a silently wrong finite value.  Weights are scaled at pack time by a power of two (exact) so
// that max|w'| lies in [2^14, 2^15): the lo parts of all but negligible weights are then normal halves; the accumulator is
// scaled back by the exact inverse in the epilogue.  Small activations whose lo part is a subnormal half lose nothing that
// matters: the absolute error of a subnormal lo is <= 2^-25.
//
// Work decomposition (512 threads = 8 waves, one workgroup per CU):
//   * a workgroup owns a 16x16 output tile of ONE image (BM = 256 pixels) x BN = 128 (or 64) output channels; waves 4 (M) x 2 (N),
//     each 64 pixels x 64 (32) channels = 2 x 2 (2 x 1) MFMA tiles of 32x32, 64 (32) accumulator registers;
//   * K loop over 32-channel slabs: the 18x18 halo patch of the slab is gathered once -- GroupNorm-apply / swish or LeakyReLU
//     prologue, channel concat and zero padding resolved in the gather exactly as in cf_igemm.hip -- split into hi / lo halves
//     and written to LDS ([pixel][hi 32 | lo 32 | pad], 144-byte rows); all nine taps reuse it;
//   * the weight slab of one (tap, K slab) -- [n][hi 32 | lo 32] halves, packed by cf_pack_conv_weight_f16x2 -- rides in a
//     3-deep LDS ring: fetched two steps ahead into registers, written one step ahead, so no wave waits on a fetch;
//   * per step (tap x 32 channels) a wave issues 24 MFMAs (768 matrix-pipe cycles) between two workgroup barriers, against 16
//     ds_read_b128 of operand fragments;
//   * nearest-x2 + 3x3 (Upsample, vqgan_arch.py:134-138) runs in the folded sub-pixel form of cf_igemm.hip (TAPS = 4: one output
//     parity class per workgroup, a 17x17 source patch, taps pre-summed at pack time);
//   * 3x3 stride 2 with a zero row / column appended bottom / right (Downsample, vqgan_arch.py:117-126) also runs as TAPS = 4: the input is
//     read as its space-to-depth view X[i][j][(p, q, c)] = x[2i + p][2j + q][c] -- two pointers (even / odd tensor rows), 2C-channel
//     "pixels", no copy -- and the conv is the 2x2 stride-1 convolution of X with the taps scattered into 4C channels at pack time
//     (SplitArgs.s2; 7 of the 16 tap x parity blocks are zero: 16/9 of the necessary MFMAs, still 1/3 of the fp32 pipe's time);
//   * epilogue as in cf_igemm.hip: per-wave LDS transpose, 16-byte stores, bias / residual / SFT, fp64 GroupNorm partials.
#include <type_traits>

#include <cstdlib>

#include "cf_common.h"

// SP_ABLATE: timing-only ablation builds (tools/split_ab.sh), a bit mask: 1 no epilogue, 2 no weight fetch, 4 no weight LDS write,
// 8 no fragment reads, 16 no per-step barrier, 32 no MFMA, 64 no activation gather, 128 no prologue/split; 0 in every product build.
#ifndef SP_FAST_RCP
#define SP_FAST_RCP 1   // 1: swish reciprocal on the raw v_rcp_f32 (1 ulp) instead of the IEEE-rounded division sequence: +4..8 %
#endif
// SP_WEAVE: 1 = fragment reads / weight fetches / weight LDS writes are woven one-by-one into the MFMA stream (sched_group_barrier)
#ifndef SP_NARROW_OCC
#define SP_NARROW_OCC 3  // workgroups per CU the 64-wide-tile instantiations are compiled for (47 KB LDS each; 168 VGPRs)
#endif
#ifndef SP_NARROW_MAX_WGS
#define SP_NARROW_MAX_WGS 64  // at most this many 128-wide workgroups per image -> use 64-wide tiles (see cf_split_launch)
#endif
#ifndef SP_WEAVE
#define SP_WEAVE 1
#endif
#ifndef SP_S2_OCC2
#define SP_S2_OCC2 1  // the 64-wide-tile instantiation of the stride-2 form is compiled for two workgroups per CU (at three it spills 20 registers)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Workgroup shapes: WM waves along M (64 pixels = 4 tile rows of 16 each) x 2 waves along N.
//   WM = 2: 256 threads, 8x16 output tile, 71 KB LDS -> two workgroups per CU (their store / gather phases overlap each other's MFMA
//           phases, and a barrier in one leaves the SIMDs to the other);   WM = 4: 512 threads, 16x16 tile, one workgroup per CU.
#ifndef SP_WM
#define SP_WM 2
#endif
constexpr int SP_KC = 32;   // channels per K slab
constexpr int SP_PW = 18;   // LDS rows per patch row (18 halo columns; the folded upsample uses 17 of them)
// LDS rows are 128 B = eight 16-byte chunks [hi k0-7 | hi k8-15 | hi k16-23 | hi k24-31 | lo k0-7 | ... | lo k24-31], UNPADDED; chunk c of
// row r is stored at position c ^ s(r), s = (x >> 1) & 7 with x = the row's patch column (A) or its channel row n (B).  A 16-lane group
// of a ds_read_b128 fragment read touches 16 consecutive x (mod 16) at one chunk c: (x & 1) picks the 128-byte half of the 256-byte
// bank window and c ^ ((x >> 1) & 7) one of its 8 slots -- 16 distinct slots, conflict-free, also for the two patch rows of a
// 32-pixel fragment (18 is even, so row parity == column parity).

struct SplitArgs {
  const float* in0;
  const float* in1;
  int c0, c1, cin, nchunks;
  int batch, hin, win, hout, wout;
  int cout, cout_pad;
  int prologue, epilogue;
  const float* pro_scale;
  const float* pro_shift;
  const float* weight;  // [class][cin/32][tap][cout_pad][hi 32 | lo 32] halves = 32 floats per row
  const float* bias;
  const float* res;
  const float* sft_scale;
  float sft_w;
  float acc_scale;  // exact power of two: 1 / (weight scale applied at pack time)
  const float* act_scale;  // [batch][2] (s, 1/s) powers of two for un-normalised inputs (prologue NONE / LEAKY), or null
  float* out;
  double* stats_out;
  int stats_cpg, nparts;
  int tiles_x, tiles_per_img, ntn;
  int nt_out;   // non-temporal output stores (cf_common.h: cf_store16)
  int s2_skip;  // stride-2 form: 1 = skip the (tap, parity) steps whose weight block is zero by construction (C % 32 == 0)
};

template <int TAPS, int NI, int WM>
struct SplitCfg {
  static constexpr int NT = WM * 128;
  static constexpr int BN = 2 * NI * 32;
  static constexpr int TH = WM * 4;                       // tile rows (16 columns)
  static constexpr int HH = TAPS == 9 ? TH + 2 : TAPS == 4 ? TH + 1 : TH;  // halo patch rows (1x1: the tile itself)
  static constexpr int HW = TAPS == 9 ? 18 : TAPS == 4 ? 17 : 16;          // halo patch columns
  static constexpr int NPIX = HH * HW;
  static constexpr int PPR = NT / 4;                      // pixels gathered per round
  static constexpr int APT = (NPIX + PPR - 1) / PPR;      // gather items (pixel, channel octet) per thread
  static constexpr int BPT = BN * 8 / NT;                 // 16-byte weight items per thread and step
  static constexpr int A_FLOATS = HH * SP_PW * 32;
  static constexpr int B_SLOT = BN * 32;
  static constexpr int LDW = NI * 32 + 4;
  static constexpr int EPI_FLOATS = WM * 2 * 32 * LDW;
  static constexpr int MAIN_FLOATS = A_FLOATS + 3 * B_SLOT;
  static constexpr int LDS_FLOATS = MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS;
};

// x -> (hi, lo) halves of two neighbouring channels, packed for the LDS rows
__device__ __forceinline__ void split2(float x0, float x1, float& hi, float& lo) {
  const f16x2 h = {(_Float16)x0, (_Float16)x1};
  const f16x2 l = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
  hi = __builtin_bit_cast(float, h);
  lo = __builtin_bit_cast(float, l);
}

// BIO (round 6, cf_conv_desc.io_bf16; TAPS == 1 only -- the ResBlock skip convolutions of the bf16-storage mode): in0, in1, res, sft_scale and out
// are bf16 tensors.  A bf16 value is its own hi half (8 significant bits, exact in an IEEE half behind the range scale) and its lo half is
// zero; the weights keep hi + lo, so the product is as exact as on fp32 tensors.  16 bytes per gather item instead of 32, 8 per output quad.
template <int TAPS, int NI, int WM, bool S2 = false, bool BIO = false>
__global__ __launch_bounds__(WM * 128, WM == 2 ? (NI == 1 && TAPS != 1 && !(S2 && SP_S2_OCC2) ? SP_NARROW_OCC : 2) : 1) void split_conv_kernel(const SplitArgs a) {
  static_assert(!S2 || TAPS == 4, "the stride-2 form is a 2x2 convolution");
  static_assert(!BIO || TAPS == 1, "bf16 tensors: the streaming 1x1 form only");
  using C = SplitCfg<TAPS, NI, WM>;
  constexpr int MI = 2;
  constexpr int NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As = smem;
  float* const Bs = smem + C::A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1;
  const int wn = wave & 1;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  int bid = blockIdx.x;
  {  // XCD-contiguous tile order (see cf_igemm.hip): neighbouring tiles share halos and weight slabs in one L2
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nt = bid % a.ntn;
  const int mt = bid / a.ntn;
  const int n0 = nt * C::BN;
  const int b = mt / a.tiles_per_img;
  int rt = mt - b * a.tiles_per_img;
  int sub_y = 0, sub_x = 0;  // TAPS == 4: output parity class of this workgroup; y0 / x0 are SOURCE coordinates
  constexpr bool s2 = S2;
  if (TAPS == 4) {
    if (s2) {
      sub_y = sub_x = 1;  // patch rows y0 .. y0 + TH, columns x0 .. x0 + 16: the geometry of parity class (1, 1)
    } else {
      // class-minor order (round 5): the four parity classes of one source tile are four CONSECUTIVE workgroups of one XCD (the tile
      // order above), i.e. they run at the same time against the same L2 -- the 9x17 source patch comes from HBM once instead of
      // once per class (counters: 1.73x the algorithmic bytes with the class-major order of rounds 2-4)
      const int cls = rt & 3;
      rt >>= 2;
      sub_y = cls >> 1;
      sub_x = cls & 1;
    }
  }
  const int tyw = rt / a.tiles_x;
  const int y0 = tyw * C::TH;
  const int x0 = (rt - tyw * a.tiles_x) * 16;

  // ---- gather geometry: item j of this thread = channel octet k8 of halo pixel p = (tid >> 2) + PPR j ----
  const int k8 = tid & 3;
  int pix[C::APT];   // source pixel index, -1 = zero padding / past the patch
  int aoff[C::APT];  // LDS float offset of the item's hi chunk (the lo chunk is at aoff ^ 16), -1 = past the patch
#pragma unroll
  for (int j = 0; j < C::APT; ++j) {
    const int p = (tid >> 2) + C::PPR * j;
    int v = -1, off = -1;
    if (p < C::NPIX) {
      const int hy = p / C::HW;
      const int hx = p - hy * C::HW;
      const int iy = y0 - (TAPS == 1 ? 0 : 1) + (TAPS == 4 ? sub_y : 0) + hy;
      const int ix = x0 - (TAPS == 1 ? 0 : 1) + (TAPS == 4 ? sub_x : 0) + hx;
      // (stride-2 form: a "pixel" is the 2C-channel pair (2 ix, 2 ix + 1) of tensor row 2 iy [in0] or 2 iy + 1 [in1 = in0 + one row])
      if (iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) v = s2 ? (b * a.hin + iy) * 2 * a.win + ix : (b * a.hin + iy) * a.win + ix;
      off = (hy * SP_PW + hx) * 32 + ((k8 ^ ((hx >> 1) & 7)) << 2);
    }
    pix[j] = v;
    aoff[j] = off;
  }

  const bool affine = a.prologue == CF_PRO_AFFINE || a.prologue == CF_PRO_AFFINE_SWISH;
  // range scale of an un-normalised input (cf_conv_desc.act_scale): powers of two, x * s and acc / s are exact; 1 when unused
  float act_s = 1.f, act_is = 1.f;
  if (!affine && a.act_scale) {
    act_s = a.act_scale[2 * b];
    act_is = a.act_scale[2 * b + 1];
  }
  const float act_s02 = 0.2f * act_s;  // LeakyReLU slope folded with the scale: fl(y * (0.2 s)) == fl(0.2 y) * s
  const float* const tab_sc = affine ? a.pro_scale + (size_t)b * a.cin : a.in0;  // (any valid address when unused)
  const float* const tab_sh = affine ? a.pro_shift + (size_t)b * a.cin : a.in0;

  // raw fp32 activations of the next slab (fetched a slab ahead), then -- converted in place -- their hi / lo halves
  struct ASet {
    f32x4 v[C::APT][2];
    f32x4 sc[2], sh[2];
  };
  ASet ra;
  // Loads are issued unconditionally from clamped addresses (a load under a divergent branch is waited for on the spot);
  // out-of-image items are zeroed at conversion time.
  auto load_A = [&](ASet& r, int chunk) __attribute__((always_inline)) {
    const int c = chunk * SP_KC;  // (uniform) first channel of the slab
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      r.sc[u] = *reinterpret_cast<const f32x4*>(tab_sc + (affine ? c + k8 * 8 + 4 * u : 0));
      r.sh[u] = *reinterpret_cast<const f32x4*>(tab_sh + (affine ? c + k8 * 8 + 4 * u : 0));
    }
    const bool first = c < a.c0;  // a slab never straddles the concat boundary (c0 % 32 == 0)
    const float* src = (first ? a.in0 : a.in1) + (first ? c : c - a.c0);
    const unsigned cs = first ? a.c0 : a.c1;
#pragma unroll
    for (int j = 0; j < C::APT; ++j) {
      const unsigned pj = pix[j] < 0 ? 0u : (unsigned)pix[j];
      if constexpr (BIO) {   // eight bf16 channels = 16 bytes, widened here (a shift)
        const cf_u32x4 q16 = *reinterpret_cast<const cf_u32x4*>(reinterpret_cast<const unsigned short*>(first ? a.in0 : a.in1) + (size_t)(pj * cs + k8 * 8) + (first ? c : c - a.c0));
        r.v[j][0] = cf_bf16x4_widen(cf_u32x2{q16[0], q16[1]});
        r.v[j][1] = cf_bf16x4_widen(cf_u32x2{q16[2], q16[3]});
        continue;
      }
      const float* q = src + (size_t)(pj * cs + k8 * 8);  // (element offsets fit 32 bits: tensors < 16 GiB)
#pragma unroll
      for (int u = 0; u < 2; ++u) r.v[j][u] = *reinterpret_cast<const f32x4*>(q + 4 * u);
    }
  };
  // prologue (GroupNorm apply / swish / LeakyReLU) + split: ra[j][0] <- 8 hi halves, ra[j][1] <- 8 lo halves.
  // Zero padding stays exactly zero: it pads the conv INPUT, i.e. the post-activation tensor.
  auto convert_mode = [&](ASet& r, auto mode) __attribute__((always_inline)) {
    constexpr int PRO = decltype(mode)::value;
#pragma unroll
    for (int j = 0; j < C::APT; ++j) {
      const bool valid = pix[j] >= 0;
      float y[8];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = r.v[j][u][e];
          if (PRO == CF_PRO_AFFINE) v = v * r.sc[u][e] + r.sh[u][e];
          if (PRO == CF_PRO_AFFINE_SWISH) {
            v = v * r.sc[u][e] + r.sh[u][e];
            v = SP_FAST_RCP ? v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)) : v * __frcp_rn(1.0f + __expf(-v));  // hardware exp / rcp swish
          }
          if (PRO == CF_PRO_LEAKY) v = v * (v > 0.f ? act_s : act_s02);
          if (PRO == CF_PRO_NONE) v = v * act_s;
          y[u * 4 + e] = valid ? v : 0.f;
        }
      f32x4 hi, lo;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        float ph, pl;
        split2(y[2 * h], y[2 * h + 1], ph, pl);
        hi[h] = ph;
        lo[h] = pl;
      }
      r.v[j][0] = hi;
      r.v[j][1] = lo;
    }
  };
  auto convert = [&](ASet& r) __attribute__((always_inline)) {
    switch (a.prologue) {
      case CF_PRO_AFFINE: convert_mode(r, std::integral_constant<int, CF_PRO_AFFINE>{}); break;
      case CF_PRO_AFFINE_SWISH: convert_mode(r, std::integral_constant<int, CF_PRO_AFFINE_SWISH>{}); break;
      case CF_PRO_LEAKY: convert_mode(r, std::integral_constant<int, CF_PRO_LEAKY>{}); break;
      default: convert_mode(r, std::integral_constant<int, CF_PRO_NONE>{}); break;
    }
  };
  auto store_A = [&](const ASet& r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < C::APT; ++j) {
      if ((j + 1) * C::PPR <= C::NPIX || aoff[j] >= 0) {
        *reinterpret_cast<f32x4*>(As + aoff[j]) = r.v[j][0];
        *reinterpret_cast<f32x4*>(As + (aoff[j] ^ 16)) = r.v[j][1];  // chunk c + 4: bit 2 of the chunk index = bit 4 of the float offset
      }
    }
  };

  // weight slabs: consecutive steps (slab-major, tap-minor) are consecutive [cout_pad][32] blocks of the packed tensor
  const size_t wstride = (size_t)a.cout_pad * 32;
  const float* const wbase = a.weight + (TAPS == 4 && !s2 ? (size_t)(sub_y * 2 + sub_x) * a.nchunks * TAPS * wstride : 0) + (size_t)n0 * 32;
  int boff[C::BPT];  // LDS float offset inside a ring slot
#pragma unroll
  for (int j = 0; j < C::BPT; ++j) {
    const int f = tid + NT * j, row = f >> 3, c = f & 7;
    boff[j] = row * 32 + ((c ^ ((row >> 1) & 7)) << 2);
  }
  f32x4 rb[C::BPT];
  auto load_B = [&](int step) __attribute__((always_inline)) {
    const float* src = wbase + (size_t)step * wstride;
#pragma unroll
    for (int j = 0; j < C::BPT; ++j) rb[j] = *reinterpret_cast<const f32x4*>(src + (tid + NT * j) * 4);
  };
  auto store_B = [&](int slot) __attribute__((always_inline)) {
    float* dst = Bs + slot * C::B_SLOT;
#pragma unroll
    for (int j = 0; j < C::BPT; ++j) *reinterpret_cast<f32x4*>(dst + boff[j]) = rb[j];
  };

  // ---- MFMA operand addresses of this lane: lane l holds row l & 31, k = (l >> 5) * 8 .. + 7 of a 16-wide K block --------------
  // chunk of (part, kk) = part*4 + kk*2 + half; swizzled position = chunk ^ s, i.e. float offset (part*16 + kk*8) ^ ((half ^ s) << 2)
  const int px = l31 & 15;
  int a_adr[3][4];  // [tap column tx][part*2 + kk]: patch pixel (py0, px + tx); mi / tap row are immediate offsets
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int hx = px + tx;
    const int hs = ((half ^ (hx >> 1)) & 7) << 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) a_adr[tx][q] = ((wm * 4 + (l31 >> 4)) * SP_PW + hx) * 32 + ((q * 8) ^ hs);
  }
  int b_adr[4];
  {
    const int hs = ((half ^ (l31 >> 1)) & 7) << 2;  // (ni*32 and wn*64 leave bits 1..3 of n alone)
#pragma unroll
    for (int q = 0; q < 4; ++q) b_adr[q] = (wn * (NI * 32) + l31) * 32 + ((q * 8) ^ hs);
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  struct Frags {
    f32x4 ah[MI], al[MI], bh[NI], bl[NI];
  };
  auto read_frags = [&](Frags& f, int tap, int slot, int kk) __attribute__((always_inline)) {  // kk: 16-wide K block of the slab
    const int ty = TAPS == 4 ? (tap >> 1) : TAPS == 1 ? 0 : tap / 3, tx = TAPS == 4 ? (tap & 1) : TAPS == 1 ? 0 : tap % 3;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f.ah[mi] = *reinterpret_cast<const f32x4*>(As + a_adr[tx][kk] + (mi * 2 + ty) * (SP_PW * 32));
      f.al[mi] = *reinterpret_cast<const f32x4*>(As + a_adr[tx][2 + kk] + (mi * 2 + ty) * (SP_PW * 32));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      f.bh[ni] = *reinterpret_cast<const f32x4*>(Bs + slot * C::B_SLOT + b_adr[kk] + ni * (32 * 32));
      f.bl[ni] = *reinterpret_cast<const f32x4*>(Bs + slot * C::B_SLOT + b_adr[2 + kk] + ni * (32 * 32));
    }
  };
  // lo*hi + hi*lo + hi*hi; consecutive MFMAs go to different accumulators
  auto mma = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.al[mi]), __builtin_bit_cast(f16x8, f.bh[ni]),
                                                             acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.ah[mi]), __builtin_bit_cast(f16x8, f.bl[ni]),
                                                             acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.ah[mi]), __builtin_bit_cast(f16x8, f.bh[ni]),
                                                             acc[mi][ni], 0, 0, 0);
  };

  // ---- software-pipelined main loop (the schedule of cf_igemm.hip) -----------------------------------------------------------
  //     step s:  [barrier] fetch B(s+2) | read frags(s, k 16..31) | 12 MFMA on frags(s, k 0..15)
  //                        read frags(s+1, k 0..15) | 12 MFMA on frags(s, k 16..31) | LDS write B(s+2)
  // RAW: B(s+1) was written before the barrier opening step s.  WAR: ring slot (s+2)%3 last held B(s-1), whose reads every wave
  // finished before the barrier opening step s.  The halo patch is single-buffered: the slab boundary (1 step in 9) drains --
  // barrier, patch write, barrier -- but the next slab's activations were fetched at tap 0 and converted (prologue + split) in
  // the shadow of the MFMAs of a middle tap, so only the LDS stores sit between the two barriers.
  const int nsteps = a.nchunks * TAPS;
  [[maybe_unused]] ASet rn;  // 1x1: the second activation set (slabs alternate between ra and rn, two slabs in flight)
  load_A(ra, 0);
  load_B(0);
  if constexpr (TAPS == 1) load_A(rn, 1 < a.nchunks ? 1 : 0);
  {
    f32x4 rb0[C::BPT];
#pragma unroll
    for (int j = 0; j < C::BPT; ++j) rb0[j] = rb[j];
    load_B(1 < nsteps ? 1 : 0);
    convert(ra);
    store_A(ra);
#pragma unroll
    for (int j = 0; j < C::BPT; ++j) *reinterpret_cast<f32x4*>(Bs + boff[j]) = rb0[j];
    store_B(1);
  }
  __syncthreads();
  Frags fx, fy;
  read_frags(fx, 0, 0, 0);
  int slot = 0;
  int step = 0;
  constexpr int CONV_TAP = TAPS == 9 ? 4 : 2;
  if constexpr (TAPS != 1)
  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    // Stride-2 form (round 5): the slab holds 32 channels of ONE parity (p, q) of the space-to-depth view (C % 32 == 0); tap (ty, tx) of
    // parity (p, q) is the 3x3 weight (2 ty + p, 2 tx + q), which does not exist for ty & p or tx & q: 7 of the 16 (tap, parity) blocks are
    // zeros by construction.  Their steps keep the pipeline's shape (weight prefetch, ring rotation, barrier) and skip the fragment reads
    // and the MFMAs: an exact zero is not added (x * 0 contributes nothing for finite x; a non-finite x still reaches the output through
    // the tap of its parity that exists).
    [[maybe_unused]] bool par_p = false, par_q = false;
    if constexpr (S2) {
      if (a.s2_skip) {
        const int c = chunk * SP_KC;
        par_p = c >= a.c0;
        par_q = (par_p ? c - a.c0 : c) >= (a.c0 >> 1);
      }
    }
    auto is_zero = [&](int tap) __attribute__((always_inline)) { return S2 && (((tap >> 1) && par_p) || ((tap & 1) && par_q)); };
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap, ++step) {
      const int slot1 = slot == 2 ? 0 : slot + 1;
      const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
      const bool zero_cur = is_zero(tap);                                // (wave-uniform)
      const bool zero_next = tap != TAPS - 1 ? is_zero(tap + 1) : false;   // (the next slab's first tap always exists)
      load_B(step + 2 < nsteps ? step + 2 : nsteps - 1);  // clamped: the tail prefetches are harmless re-reads
      if (tap == 0) load_A(ra, chunk + 1 < a.nchunks ? chunk + 1 : chunk);
      if (!zero_cur) read_frags(fy, tap, slot, 1);
      constexpr int NMF = MI * NI * 3;  // MFMAs per half step
      const bool weave = SP_WEAVE && !S2 && NMF >= 2 * (MI + NI) + C::BPT && tap != CONV_TAP && tap != 0;   // (the stride-2 form branches around its MFMA blocks: no weave)
      if (!weave) __builtin_amdgcn_sched_barrier(0);  // pin the fetches above the MFMA block (hipcc would sink them next to their use)
      if (!zero_cur) mma(fx);
      if (tap == CONV_TAP) convert(ra);
      if (weave) {
        // MFMA-first: frags(s, k 0..15) are in registers, so the half step opens with an MFMA and every LDS read / fetch issues in
        // the shadow of one
#pragma unroll
        for (int i = 0; i < C::BPT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
        }
#pragma unroll
        for (int i = 0; i < 2 * (MI + NI); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        }
#pragma unroll
        for (int i = 0; i < NMF - 2 * (MI + NI) - C::BPT; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tap != TAPS - 1 && !zero_next) read_frags(fx, tap + 1, slot1, 0);
      if (!zero_cur) mma(fy);
      if (!weave) __builtin_amdgcn_sched_barrier(0);
      store_B(slot2);
      if (weave) {
#pragma unroll
        for (int i = 0; i < (tap != TAPS - 1 ? 2 * (MI + NI) : 0); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < NMF - (tap != TAPS - 1 ? 2 * (MI + NI) : 0) - C::BPT; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int i = 0; i < C::BPT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write (waits for the slab fetched in the first half)
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (tap == TAPS - 1 && chunk + 1 < a.nchunks) {
        store_A(ra);
        __syncthreads();
        read_frags(fx, 0, slot1, 0);
      }
      slot = slot1;
    }
  }
  if constexpr (TAPS == 1) {
    // 1x1 (the ResBlock skip convolutions on large images: HBM-bound streaming): one step per 32-channel slab -- 12 NI MFMAs between
    // the patch rewrite -- so the activations of TWO slabs are in flight: slab c + 2 is requested at the top of step c into the set
    // slab c left, slab c + 1 (requested a step ago) is converted in the shadow of step c's MFMAs and written behind its barrier.
    auto step1 = [&](int chunk, ASet& rl, ASet& rc) __attribute__((always_inline)) {
      const int slot1 = slot == 2 ? 0 : slot + 1;
      const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
      load_B(chunk + 2 < nsteps ? chunk + 2 : nsteps - 1);
      load_A(rl, chunk + 2 < a.nchunks ? chunk + 2 : a.nchunks - 1);
      read_frags(fy, 0, slot, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(fx);
      if (chunk + 1 < a.nchunks) convert(rc);
      __builtin_amdgcn_sched_barrier(0);
      mma(fy);
      __builtin_amdgcn_sched_barrier(0);
      store_B(slot2);
      __syncthreads();
      if (chunk + 1 < a.nchunks) {
        store_A(rc);
        __syncthreads();
        read_frags(fx, 0, slot1, 0);
      }
      slot = slot1;
    };
    for (int chunk = 0; chunk < a.nchunks; chunk += 2) {
      step1(chunk, ra, rn);
      if (chunk + 1 < a.nchunks) step1(chunk + 1, rn, ra);
    }
  }

  // ---- epilogue (cf_igemm.hip's): per-wave transpose through LDS, 32 rows at a time, 16-byte accesses -----------------------
  // (the loop's closing barrier retired every LDS read of the main loop)
  constexpr int LDW = C::LDW;
  constexpr int Q = NI * 8;   // float4 per tile row
  constexpr int RPP = 64 / Q;  // rows per pass
  constexpr int PASSES = 32 / RPP;
  auto epilogue = [&](auto mode) __attribute__((always_inline)) {
    constexpr int EPI = decltype(mode)::value;
    float* stage = smem + wave * (32 * LDW);
    const int cq = lane % Q, rl = lane / Q;
    const int n = n0 + wn * (NI * 32) + cq * 4;  // (always < cout: the host requires cout % 64 == 0)
    constexpr bool nvalid = true;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n);
    const float s = a.acc_scale * act_is;  // (a product of powers of two: exact)
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    // Residual / SFT operands of BOTH 32-row halves are requested up front (the main loop's registers are free by now), so their
    // HBM latency overlaps the transposes instead of opening each half.
    unsigned offs[MI][PASSES];  // element offsets fit 32 bits (tensors < 16 GiB)
    f32x4 r0[MI][PASSES], r1[MI][PASSES];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int row = wm * 64 + mi * 32 + p * RPP + rl;
        unsigned pixel;
        if (TAPS == 4 && !s2)
          pixel = ((unsigned)b * a.hout + (2 * (y0 + (row >> 4)) + sub_y)) * a.wout + (2 * (x0 + (row & 15)) + sub_x);
        else
          pixel = ((unsigned)b * a.hout + (y0 + (row >> 4))) * a.wout + (x0 + (row & 15));
        offs[mi][p] = pixel * (unsigned)a.cout + n;
        r0[mi][p] = r1[mi][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BIO) {
          if (EPI == CF_EPI_RESIDUAL || EPI == CF_EPI_SFT) r0[mi][p] = cf_load4_bf16(a.res, offs[mi][p]);
          if (EPI == CF_EPI_SFT) r1[mi][p] = cf_load4_bf16(a.sft_scale, offs[mi][p]);
        } else {
          if (EPI == CF_EPI_RESIDUAL || EPI == CF_EPI_SFT) r0[mi][p] = *reinterpret_cast<const f32x4*>(a.res + offs[mi][p]);
          if (EPI == CF_EPI_SFT) r1[mi][p] = *reinterpret_cast<const f32x4*>(a.sft_scale + offs[mi][p]);
        }
      }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[cf_acc_row(r, lane) * LDW + ni * 32 + l31] = acc[mi][ni][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      f32x4 t[PASSES];
#pragma unroll
      for (int p = 0; p < PASSES; ++p) t[p] = *reinterpret_cast<const f32x4*>(stage + (p * RPP + rl) * LDW + cq * 4);
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        f32x4 v = t[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * s + bias4[e];
        if (EPI == CF_EPI_RESIDUAL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r0[mi][p][e];
        } else if (EPI == CF_EPI_SFT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = r0[mi][p][e] + a.sft_w * (r0[mi][p][e] * r1[mi][p][e] + v[e]);
        }
        t[p] = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ssum[e] += v[e];
          ssq[e] += v[e] * v[e];
        }
      }
      // (one wave-uniform choice of the stores' cache policy per 32-row block, not per store: cf_common.h cf_store16, cf_wf43.hip)
      if constexpr (BIO) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) cf_store4_bf16(a.out, offs[mi][p], t[p]);   // (rounded here, once)
      } else if (a.nt_out) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) __builtin_nontemporal_store(t[p], reinterpret_cast<f32x4*>(a.out + offs[mi][p]));
      } else {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) *reinterpret_cast<f32x4*>(a.out + offs[mi][p]) = t[p];
      }
      __builtin_amdgcn_wave_barrier();  // the staging rows are rewritten by the next 32-row block
    }
    if (a.stats_out) {
      // GroupNorm statistics of the values just written (fp64 partials, fixed shuffle order): one partial per
      // (image, group, tile, wave row) -- nparts = tiles_per_img * WM
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
        const size_t pidx = (size_t)(mt - b * a.tiles_per_img) * WM + wm;
        const int ng = a.cout / cpg;
        double* o = a.stats_out + (((size_t)b * ng + n / cpg) * a.nparts + pidx) * 2;
        o[0] = d0;
        o[1] = q0;
        if (cpg == 2) {
          o[(size_t)a.nparts * 2] = d1;  // group n/2 + 1
          o[(size_t)a.nparts * 2 + 1] = q1;
        }
      }
    }
  };
  switch (a.epilogue) {
    case CF_EPI_RESIDUAL: epilogue(std::integral_constant<int, CF_EPI_RESIDUAL>{}); break;
    case CF_EPI_SFT: epilogue(std::integral_constant<int, CF_EPI_SFT>{}); break;
    default: epilogue(std::integral_constant<int, CF_EPI_NONE>{}); break;
  }
}

// packed slab entry (cf_igemm.hip's rule): plain tap, or the pre-summed taps of the folded nearest-x2 + 3x3
__device__ __forceinline__ float split_weight_value(const float* __restrict__ w, int cout, int cin, int fold, int slab, int n, int c) {
  if (n >= cout || c >= cin) return 0.f;
  const float* wk = w + ((long)n * cin + c) * 9;
  if (fold == 3) return w[(long)n * cin + c];  // 1x1 weight [cout][cin]
  if (!fold) return wk[slab];
  if (fold == 2) return 0.f;  // (stride-2 form: handled by split_weight_value_s2)
  const int cls = slab >> 2, t2 = slab & 3;
  const int sy = cls >> 1, sx = cls & 1, ty = t2 >> 1, tx = t2 & 1;
  const int ky0 = sy == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = sy == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
  const int kx0 = sx == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = sx == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
  float v = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) v += wk[ky * 3 + kx];
  return v;
}

// Stride-2 form (Downsample: zero row / column appended bottom / right, 3x3 stride 2 -- vqgan_arch.py:117-126): the input is read as the
// space-to-depth tensor X[i][j][(p, q, c)] = x[2i + p][2j + q][c] (4C channels, no copy: see the row-pair addressing of the gather) and
// the convolution becomes a 2x2 stride-1 one, out[i][j] = sum_{ty,tx} W'[ty][tx] . X[i + ty][j + tx], with
// W'[ty][tx][(p, q, c)] = w[2ty + p][2tx + q][c] where that tap exists and 0 elsewhere (7 of the 16 blocks are zero).
__device__ __forceinline__ float split_weight_value_s2(const float* __restrict__ w, int cout, int cin, int tap, int n, int c4) {
  if (n >= cout || c4 >= 4 * cin) return 0.f;
  const int p = c4 / (2 * cin), q = (c4 / cin) & 1, c = c4 % cin;
  const int ky = 2 * (tap >> 1) + p, kx = 2 * (tap & 1) + q;
  if (ky > 2 || kx > 2) return 0.f;
  return w[((long)n * cin + c) * 9 + ky * 3 + kx];
}

// [class][cin/32][tap][cout_pad][32 words] (class = 1 plain / 4 folded / 1 stride-2 form with 4 cin channels): words 0..15 = hi halves of channels (2k, 2k+1), 16..31 = lo
__global__ void pack_weight_f16x2_kernel(const float* __restrict__ w, int cout, int cin, int fold, int cout_pad, int nchunks,
                                         float scale, unsigned* __restrict__ packed, long total_words) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total_words) return;
  const int k2 = (int)(i & 15), part = (int)((i >> 4) & 1);
  long r = i >> 5;
  const int n = (int)(r % cout_pad);
  r /= cout_pad;
  const int taps = fold == 3 ? 1 : fold ? 4 : 9;
  const int tap = (int)(r % taps);
  r /= taps;
  const int chunk = (int)(r % nchunks);
  const int slab = (int)(r / nchunks) * 4 + tap;  // class * 4 + tap (class = 0 when not folded)
  unsigned out = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = chunk * 32 + k2 * 2 + h;
    const float v = (fold == 2 ? split_weight_value_s2(w, cout, cin, tap, n, c) : split_weight_value(w, cout, cin, fold, slab, n, c)) * scale;  // exact: power of two
    const _Float16 hi = (_Float16)v;
    const _Float16 hv = part ? (_Float16)(v - (float)hi) : hi;
    out |= (unsigned)__builtin_bit_cast(unsigned short, hv) << (16 * h);
  }
  packed[i] = out;
}

template <int TAPS, int NI, bool S2 = false, bool BIO = false>
int split_launch(SplitArgs& k, int batch, hipStream_t stream) {
  using C = SplitCfg<TAPS, NI, SP_WM>;
  k.ntn = k.cout_pad / C::BN;
  constexpr auto kern = split_conv_kernel<TAPS, NI, SP_WM, S2, BIO>;
  constexpr size_t lds = C::LDS_FLOATS * sizeof(float);
  CF_LDS_ATTR(kern, lds);  // (cf_device_init sets the dynamic-LDS attribute on each device)
  hipLaunchKernelGGL(kern, dim3(k.tiles_per_img * batch * k.ntn), dim3(C::NT), lds, stream, k);
  CF_CHECK_LAUNCH("cf_conv2d(f16x2)");
  return CF_OK;
}

}  // namespace

extern "C" int cf_pack_conv_weight_f16x2(const float* w, int cout, int cin, int up2x, int cout_pad, int cin_pad, float scale,
                                         void* packed, cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_f16x2: null pointer");
  CF_REQUIRE((up2x == 2 ? 4 * cin_pad : cin_pad) % 32 == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 64 == 0,
             "cf_pack_conv_weight_f16x2: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  int ex = 0;
  CF_REQUIRE(scale > 0.f && frexpf(scale, &ex) == 0.5f, "cf_pack_conv_weight_f16x2: scale %g is not a power of two", (double)scale);
  CF_REQUIRE(up2x >= 0 && up2x <= 3, "cf_pack_conv_weight_f16x2: form %d (0 plain 3x3, 1 nearest-x2 folded, 2 stride 2, 3 1x1)", up2x);
  CF_REQUIRE(up2x != 2 || (cin_pad == cin && cin % 16 == 0), "cf_pack_conv_weight_f16x2: the stride-2 form needs cin %% 16 == 0, unpadded");
  const long words = (long)(up2x == 3 ? 1 : up2x ? 16 : 9) * cin_pad * cout_pad;  // two halves per word, hi + lo per channel: one word per weight
  hipLaunchKernelGGL(pack_weight_f16x2_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                     up2x, cout_pad, (up2x == 2 ? 4 * cin_pad : cin_pad) / 32, scale, reinterpret_cast<unsigned*>(packed), words);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_f16x2");
  return CF_OK;
}

// Called by cf_conv2d (cf_igemm.hip) for descriptors with bf16_mfma == CF_OPERAND_F16X2; the common argument checks have run.
int cf_split_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query) {
  const bool s2 = d->stride == 2;
  const bool one = d->taps == 1;
  CF_REQUIRE((d->taps == 9 || one) && (d->stride == 1 || s2) && !d->in_nchw && !d->out_nchw && !d->winograd,
             "cf_conv2d: f16x2 operands cover 3x3 NHWC convolutions (stride 1 plain or nearest-x2 folded, stride 2) and 1x1 on images");
  CF_REQUIRE(!one || (d->stride == 1 && !d->upsample && !d->stats_out), "cf_conv2d(f16x2): 1x1 convolutions are stride 1, no upsample, no statistics");
  CF_REQUIRE(!d->io_bf16 || one, "cf_conv2d(f16x2): bf16 tensors (io_bf16) are read by the streaming 1x1 form only");
  if (s2)
    CF_REQUIRE(!d->upsample && d->c1 == 0 && d->c0 % 16 == 0 && d->pad_lo == 0 && d->hin % 2 == 0 && d->win % 2 == 0,
               "cf_conv2d(f16x2): stride 2 needs one dense input with c0 %% 16 == 0 (got %d), even size, padding bottom / right", d->c0);
  else
    CF_REQUIRE(d->c0 % 32 == 0 && d->c1 % 32 == 0, "cf_conv2d(f16x2): input channels (%d, %d) must be multiples of 32", d->c0, d->c1);
  CF_REQUIRE(d->cout % 64 == 0 && d->cout_pad == d->cout, "cf_conv2d(f16x2): cout %d / cout_pad %d must be one multiple of 64", d->cout,
             d->cout_pad);
  constexpr int TH = SP_WM * 4;
  const int gh = s2 ? d->hin / 2 : d->hin, gw = s2 ? d->win / 2 : d->win;  // the grid the tiles live on
  CF_REQUIRE(gh % TH == 0 && gw % 16 == 0, "cf_conv2d(f16x2): %dx%d %s is not a multiple of the %dx16 tile", gh, gw, s2 ? "output" : "input", TH);
  CF_REQUIRE(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_RESIDUAL || d->epilogue == CF_EPI_SFT,
             "cf_conv2d(f16x2): epilogues are none / residual / SFT");
  CF_REQUIRE(d->pad_mode == CF_PAD_ZERO && (d->ld_in0 == 0 || d->ld_in0 == d->c0) && (d->ld_in1 == 0 || d->ld_in1 == d->c1) &&
                 (d->ld_out == 0 || d->ld_out == d->cout),
             "cf_conv2d(f16x2): dense tensors with zero padding only");
  CF_REQUIRE(d->acc_scale > 0.f, "cf_conv2d(f16x2): acc_scale must be the inverse of the pack-time weight scale (got %g)",
             (double)d->acc_scale);
  SplitArgs a;
  a.in0 = d->in0;
  a.in1 = d->in1;
  a.c0 = d->c0;
  a.c1 = d->c1;
  a.hin = d->hin;
  a.win = d->win;
  if (s2) {  // space-to-depth view: even tensor rows through in0, odd rows through in1, 2C channels (two pixels) each
    a.in1 = d->in0 + (size_t)d->win * d->c0;
    a.c0 = a.c1 = 2 * d->c0;
    a.hin = gh;
    a.win = gw;
  }
  a.cin = a.c0 + a.c1;
  a.nchunks = a.cin / SP_KC;
  a.batch = d->batch;
  a.hout = d->hout;
  a.wout = d->wout;
  a.cout = d->cout;
  a.cout_pad = d->cout_pad;
  a.prologue = d->prologue;
  a.epilogue = d->epilogue;
  a.pro_scale = d->pro_scale;
  a.pro_shift = d->pro_shift;
  a.weight = d->weight;
  a.bias = d->bias;
  a.res = d->res;
  a.sft_scale = d->sft_scale;
  a.sft_w = d->sft_w;
  a.acc_scale = d->acc_scale;
  a.act_scale = d->act_scale;
  a.out = d->out;
  a.stats_out = d->stats_out;
  a.stats_cpg = d->stats_cpg > 0 ? d->stats_cpg : 1;
  a.tiles_x = gw / 16;  // tiles live on the SOURCE grid (== the output grid unless upsample; stride 2: the space-to-depth grid)
  a.tiles_per_img = (d->upsample ? 4 : 1) * a.tiles_x * (gh / TH);
  a.nparts = a.tiles_per_img * SP_WM;
  a.ntn = 0;
  static const int s2_skip_on = [] {   // A/B only: CF_S2_SKIP=0 multiplies the zero blocks as rounds 3-4 did
    const char* e = getenv("CF_S2_SKIP");
    return e ? atoi(e) : 1;
  }();
  a.s2_skip = s2 && s2_skip_on && d->c0 % 32 == 0;
  a.nt_out = cf_nt_store((long)d->batch * d->hout * d->wout * d->cout * 4);
  if (parts_query) {
    *parts_query = a.nparts;
    return CF_OK;
  }
  // 128-wide channel tiles unless the layer is small (at most SP_NARROW_MAX_WGS 128-wide workgroups PER IMAGE: the 32x32 / 64x64
  // layers): 64-wide tiles double the workgroup count of a small batch.  The rule looks at the per-image shape only -- the two tile
  // widths write the same outputs but group the GroupNorm partial sums differently, so a batch-dependent choice would break the
  // bitwise batch invariance of the network.
  static const int narrow_max_wgs = [] {
    const char* e = getenv("CODEFORMER_HIP_SPLIT_NARROW_WGS");
    return e ? atoi(e) : SP_NARROW_MAX_WGS;
  }();
  const bool wide = d->cout_pad % 128 == 0 && (long)a.tiles_per_img * (d->cout_pad / 128) > narrow_max_wgs;
  if (s2) return wide ? split_launch<4, 2, true>(a, d->batch, stream) : split_launch<4, 1, true>(a, d->batch, stream);
  if (one && d->io_bf16) return wide ? split_launch<1, 2, false, true>(a, d->batch, stream) : split_launch<1, 1, false, true>(a, d->batch, stream);
  if (one) return wide ? split_launch<1, 2>(a, d->batch, stream) : split_launch<1, 1>(a, d->batch, stream);
  if (d->upsample) return wide ? split_launch<4, 2>(a, d->batch, stream) : split_launch<4, 1>(a, d->batch, stream);
  return wide ? split_launch<9, 2>(a, d->batch, stream) : split_launch<9, 1>(a, d->batch, stream);
}

int cf_split_splitk_geometry(const cf_conv_desc* d, int* tiles, long* bytes_per_part) {
  (void)d;
  *tiles = 0;
  *bytes_per_part = 0;
  cf_set_error("cf_conv2d: split_k is not available for this kernel");
  return CF_ERR_ARG;
}
