// Winograd F(4x4,3x3) convolution with split-half operands, 64 output channels per eight-wave workgroup (gfx950).
//
// Serves the 3x3 stride-1 convolutions of the GENERATOR whose output has 64 channels (vqgan_arch.py:300-316: the two ResBlocks at
// 512x512) -- never the encoder, which decides the code indices and stays on F(2x2,3x3).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d: 6x6 input tile, g: 3x3 kernel, Y: 4x4 outputs
// with the interpolation points (0, +-1/2, +-2, inf) -- the set with the smallest mean error among those tools/winograd_f43_numerics.py
// compares -- and the rows of B^T scaled by D = diag(1/4, 1/4, 1/4, 1/2, 1/2, 1/4) (G by D^-1; powers of two, so every intermediate is
// the unscaled one times a power of two and rounds identically).  With D the largest absolute row sum of B^T is 1.875: a
// transform-domain value is at most 3.52 max|d|, below the 4 max|d| of F(2,3), so the IEEE-half operand range rules of the F(2,3)
// kernels (GroupNorm bound on the host, act_scale for un-normalised inputs, growth 4) hold unchanged.
//   B'^T = [ .25 0 -1.0625 0 .25 0 ;  0 -.5 -1 .125 .25 0 ;  0 .5 -1 -.125 .25 0 ;  0 -.25 -.125 1 .5 0 ;  0 .25 -.125 -1 .5 0 ;
//            0 .25 0 -1.0625 0 .25 ]
//   G'   = [ 4 0 0 ; -32/15 -16/15 -8/15 ; -32/15 16/15 -8/15 ; 1/15 2/15 4/15 ; 1/15 -2/15 4/15 ; 0 0 4 ]
//   A^T  = [ 1 1 1 1 1 0 ; 0 .5 -.5 2 -2 0 ; 0 .25 .25 4 4 0 ; 0 .125 -.125 8 -8 1 ]
// The 36 transform-domain GEMMs  M[xi,nu][tile][n] = sum_c V[xi,nu][tile][c] U[xi,nu][c][n]  run on v_mfma_f32_32x32x16_f16 with U and V
// as hi + lo IEEE halves (hi*hi + lo*hi + hi*lo, fp32 accumulation), as in cf_wsplit.hip: 36 positions per 16 outputs = 2.25 products per
// output pixel and input channel instead of 4 -- and with them 0.56 of the MFMAs, of the weight-fragment bytes through the
// vector-memory path, of the V bytes through LDS and of the operand splits per output; the halo factor drops from 1.41 to 1.20.
//
// Why 64 output channels and not 128: one MFMA row tile is 32 Winograd tiles = a 16x32-pixel patch, and the accumulators of 36
// positions x 32 tiles x 128 channels (590 KB) exceed the CU's 512 KB register file; x 64 channels they are 295 KB = 144 registers
// per lane of eight waves.  A 64-channel workgroup on a 128-channel layer repeats gather + prologue + transform per channel half,
// which costs about what the matrix side saves -- so the kernel is used where one workgroup covers the layer.
//
// Work decomposition (512 threads = 8 waves, one workgroup per CU, 148 KB LDS):
//   * a workgroup owns a 16x32 output patch of ONE image (4x8 tiles of 4x4 outputs) x 64 channels; K loop over 16-channel slabs;
//   * gather: the 18x34 halo patch of the slab (612 pixels x 4 channel quads, five float4 items per thread) is requested a whole slab
//     ahead (registers), passes the GroupNorm-apply / swish or LeakyReLU prologue, zero padding and concat as in the other kernels,
//     and is written to LDS ([pixel][20 floats]);
//   * input transform: item = (tile, channel pair, xi half): column pass for three rows of B'^T d (five of the six tile rows are read),
//     row pass for their six nu, split into hi + lo and written to V[36][32 tiles][16 hi halves | 16 lo halves] (80-byte rows);
//   * MFMA stage: wave = (xi half, nu half, channel half): nine positions x 32 channels = 144 accumulator registers; A fragments from
//     V, B fragments global/L2 -> registers through a four-position ring, MFMAs of two positions interleaved;
//   * two barriers per slab: { MMA(s) ; prologue + store(s+1) } | barrier | transform(s+1) | barrier | ...
//   * epilogue: the accumulators go through LDS in four passes (channel half x 16 tiles, [36][16 tiles][32 ch]); item = (output row a,
//     tile, channel quad) contracts xi then nu, applies acc_scale / bias / residual / SFT, stores four pixels as float4s and accumulates
//     the GroupNorm statistics of what it wrote (fp64 partials, fixed shuffle order: eight partials per patch and group).
#include <type_traits>

#include "cf_common.h"

#ifndef F4_ABLATE   // timing-only ablation builds: 1 no MFMAs, 2 no transform, 4 no prologue + store, 8 no epilogue, 16 no weight fetch
#define F4_ABLATE 0
#endif

namespace {

constexpr int F4_TH = 16, F4_TW = 32;            // output patch of a workgroup
constexpr int F4_PW = F4_TW + 2;                 // halo patch 18 x 34
constexpr int F4_NPIX = (F4_TH + 2) * F4_PW;     // 612
constexpr int F4_NT = 32;                        // tiles per patch (4 rows x 8 columns) = one MFMA row tile
constexpr int F4_THREADS = 512;
constexpr int F4_BN = 64;                        // output channels per workgroup
constexpr int F4_APT = 5;                        // float4 gather items per thread: 640 pixel slots x 4 quads / 512 threads
constexpr int F4_PATCH_FLOATS = 640 * CF_LDK;    // 612 halo pixels padded to five gather rounds of 128 pixels (no guard on the store)
constexpr int F4_PS = F4_NT * CF_LDK + 4;        // 644 floats between positions of V (see cf_winograd.hip)
constexpr int F4_V_FLOATS = 36 * F4_PS;          // 23184
constexpr int F4_RLD = 36;                       // epilogue staging row: 32 channels + 4 pad
constexpr int F4_M_FLOATS = 36 * 16 * F4_RLD;    // 20736: one pass of the epilogue (36 positions x 16 tiles x 32 channels)
constexpr int F4_TAB = 512;                      // GroupNorm scale / shift rows of the image (cin <= 512)
constexpr int F4_LDS_FLOATS = F4_PATCH_FLOATS + F4_V_FLOATS + 2 * F4_TAB;   // 148,032 bytes
static_assert(F4_M_FLOATS <= F4_V_FLOATS, "epilogue staging must fit the V buffer");

typedef _Float16 f4_f16x8 __attribute__((ext_vector_type(8)));
typedef float f4_f32x2 __attribute__((ext_vector_type(2)));

struct F4Args {
  const float* in0;
  const float* in1;
  int c0, c1, cin, nchunks;
  int batch, h, w;
  int cout;
  int prologue, epilogue;
  const float* pro_scale;
  const float* pro_shift;
  const float* weight;  // [36 pos][nchunks][cout/32][hi, lo][64 lanes][4 words]  (cf_pack_conv_weight_winograd43_f16x2)
  const float* bias;
  const float* res;
  const float* sft_scale;
  float sft_w;
  float acc_scale;
  const float* act_scale;
  float* out;
  double* stats_out;
  int stats_cpg, nparts;
  int tiles_x, tiles_per_img, ntn;
};

template <int PRO>
__global__ __launch_bounds__(F4_THREADS, 1) void wf43_kernel(const F4Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const patch = smem;
  float* const V = smem + F4_PATCH_FLOATS;
  float* const tab = V + F4_V_FLOATS;  // [scale: F4_TAB][shift: F4_TAB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform by construction: keeps what depends on it in SGPRs)
  const int half = lane >> 5;
  const int l31 = lane & 31;

  int bid = blockIdx.x;
  {  // XCD-contiguous tile order (see cf_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ntile = bid % a.ntn;
  const int mt = bid / a.ntn;
  const int n0 = ntile * F4_BN;
  const int b = mt / a.tiles_per_img;
  const int rt = mt - b * a.tiles_per_img;
  const int tyw = rt / a.tiles_x;
  const int y0 = tyw * F4_TH;
  const int x0 = (rt - tyw * a.tiles_x) * F4_TW;
  const int n = a.nchunks;

  constexpr bool affine = PRO == CF_PRO_AFFINE || PRO == CF_PRO_AFFINE_SWISH;
  if (affine) {  // this image's GroupNorm rows -> LDS, read per slab by the patch store (first use is behind the first barrier)
    for (int i = tid; i < a.cin; i += F4_THREADS) {
      tab[i] = a.pro_scale[(size_t)b * a.cin + i];
      tab[F4_TAB + i] = a.pro_shift[(size_t)b * a.cin + i];
    }
  }
  float act_s = 1.f, act_is = 1.f;
  if (!affine && a.act_scale) {
    act_s = a.act_scale[2 * b];
    act_is = a.act_scale[2 * b + 1];
  }
  const float act_s02 = 0.2f * act_s;  // LeakyReLU slope folded with the scale: fl(y * (0.2 s)) == fl(0.2 y) * s

  // ---- gather: item j of this thread is float4 #k4 of halo pixel p = (tid >> 2) + 128 j ----
  const int k4 = tid & 3;
  int pix[F4_APT];
#pragma unroll
  for (int j = 0; j < F4_APT; ++j) {
    const int p = (tid >> 2) + 128 * j;
    int v = -1;
    if (p < F4_NPIX) {
      const int hy = p / F4_PW;
      const int hx = p - hy * F4_PW;
      const int iy = y0 - 1 + hy;
      const int ix = x0 - 1 + hx;
      if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) v = (b * a.h + iy) * a.w + ix;
    }
    pix[j] = v;
  }
  f32x4 ra[F4_APT];
  // unconditional loads from clamped addresses; out-of-image items are zeroed at the store (see cf_winograd.hip)
  auto load_A = [&](int chunk) __attribute__((always_inline)) {
    const int c = chunk * CF_BK + k4 * 4;
    const bool first = c < a.c0;
    const float* src = first ? a.in0 : a.in1;
    const int cs = first ? a.c0 : a.c1;
    const int cc = first ? c : c - a.c0;
#pragma unroll
    for (int j = 0; j < F4_APT; ++j) {
      const int pj = pix[j] < 0 ? 0 : pix[j];
      ra[j] = *reinterpret_cast<const f32x4*>(src + (size_t)pj * cs + cc);
    }
  };
  auto store_patch = [&](int chunk) __attribute__((always_inline)) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
      sc = *reinterpret_cast<const f32x4*>(tab + chunk * CF_BK + k4 * 4);
      sh = *reinterpret_cast<const f32x4*>(tab + F4_TAB + chunk * CF_BK + k4 * 4);
    }
#pragma unroll
    for (int j = 0; j < F4_APT; ++j) {
      const int p = (tid >> 2) + 128 * j;  // (p >= 612: padding rows of the patch buffer, written as zeros -- no branch)
      const bool valid = pix[j] >= 0;
      f32x4 v = ra[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = v[e];
        if (PRO == CF_PRO_AFFINE) y = y * sc[e] + sh[e];
        if (PRO == CF_PRO_AFFINE_SWISH) {
          y = y * sc[e] + sh[e];
          y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));  // same hardware exp / rcp swish as the other conv kernels
        }
        if (PRO == CF_PRO_LEAKY) y = y * (y > 0.f ? act_s : act_s02);
        if (PRO == CF_PRO_NONE) y = y * act_s;
        v[e] = valid ? y : 0.f;
      }
      *reinterpret_cast<f32x4*>(patch + p * CF_LDK + k4 * 4) = v;
    }
  };

  // ---- input transform: item = (tile (ty, tx), channel pair cp, xi half th): V[(3 th + aa) * 6 + nu] for aa = 0..2, nu = 0..5 ----
  const int t_tx = lane >> 3, t_cp = lane & 7, t_ty = wave & 3, t_h = wave >> 2;
  const float* const t_in = patch + ((4 * t_ty + t_h) * F4_PW + 4 * t_tx) * CF_LDK + t_cp * 2;  // tile row i0 = th, tile column 0
  float* const t_out = V + (t_h * 18) * F4_PS + (t_ty * 8 + t_tx) * CF_LDK + t_cp;              // position (3 th, 0): hi word; lo: + 8
  // Two groups keep the live set small (the accumulators hold 144 of the 256 registers): first the single row of the half (xi 0 or 5: tile
  // rows i0, i0 + 2, i0 + 4), then the even / odd pair (xi 1, 2 or 3, 4: tile rows 1..4).  `mid` runs between the column and the row pass of
  // the second group (the slab's first weight fragments are requested there: fewest live registers).
  auto row_pass = [&](const f4_f32x2 (&zz)[6], int pos) __attribute__((always_inline)) {
    f4_f32x2 v[6];
    v[0] = (zz[0] + zz[4]) * 0.25f - zz[2] * 1.0625f;
    const f4_f32x2 e1 = zz[4] * 0.25f - zz[2];
    const f4_f32x2 o1 = zz[3] * 0.125f - zz[1] * 0.5f;
    v[1] = e1 + o1;
    v[2] = e1 - o1;
    const f4_f32x2 e2 = zz[4] * 0.5f - zz[2] * 0.125f;
    const f4_f32x2 o2 = zz[3] - zz[1] * 0.25f;
    v[3] = e2 + o2;
    v[4] = e2 - o2;
    v[5] = (zz[1] + zz[5]) * 0.25f - zz[3] * 1.0625f;
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) {  // operand split, store: [hi: 16 halves | lo: 16 halves] per (position, tile)
      float hi, lo;
      cf_split_pair(v[nu][0], v[nu][1], hi, lo);
      t_out[(pos + nu) * F4_PS] = hi;
      t_out[(pos + nu) * F4_PS + 8] = lo;
    }
  };
  auto transform = [&](auto mid) __attribute__((always_inline)) {
    {  // xi = 0 (half 0: tile rows 0, 2, 4) or xi = 5 (half 1: tile rows 1, 3, 5; t_in starts at row 1): .25 (d_a + d_c) - 1.0625 d_b
      f4_f32x2 z[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f4_f32x2 da = *reinterpret_cast<const f4_f32x2*>(t_in + (0 * F4_PW + j) * CF_LDK);
        const f4_f32x2 db = *reinterpret_cast<const f4_f32x2*>(t_in + (2 * F4_PW + j) * CF_LDK);
        const f4_f32x2 dc = *reinterpret_cast<const f4_f32x2*>(t_in + (4 * F4_PW + j) * CF_LDK);
        z[j] = (da + dc) * 0.25f - db * 1.0625f;
      }
      __builtin_amdgcn_sched_barrier(0);
      row_pass(z, t_h == 0 ? 0 : 12);  // position (3 th + aa) * 6 relative to t_out's (3 th) * 6: aa = 0 (xi 0) or aa = 2 (xi 5)
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // the even / odd pair from tile rows 1..4: half 0: e = .25 d4 - d2, o = .125 d3 - .5 d1 (xi 1, 2); half 1: e = .5 d4 - .125 d2,
       // o = d3 - .25 d1 (xi 3, 4)
      const float* tp = t_in + (1 - t_h) * F4_PW * CF_LDK;  // tile row 1
      const float ce4 = t_h == 0 ? 0.25f : 0.5f, ce2 = t_h == 0 ? 1.f : 0.125f, co3 = t_h == 0 ? 0.125f : 1.f, co1 = t_h == 0 ? 0.5f : 0.25f;
      f4_f32x2 zp[6], zm[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f4_f32x2 d1 = *reinterpret_cast<const f4_f32x2*>(tp + (0 * F4_PW + j) * CF_LDK);
        const f4_f32x2 d2 = *reinterpret_cast<const f4_f32x2*>(tp + (1 * F4_PW + j) * CF_LDK);
        const f4_f32x2 d3 = *reinterpret_cast<const f4_f32x2*>(tp + (2 * F4_PW + j) * CF_LDK);
        const f4_f32x2 d4 = *reinterpret_cast<const f4_f32x2*>(tp + (3 * F4_PW + j) * CF_LDK);
        const f4_f32x2 e = d4 * ce4 - d2 * ce2;
        const f4_f32x2 o = d3 * co3 - d1 * co1;
        zp[j] = e + o;
        zm[j] = e - o;
        if (j == 2) __builtin_amdgcn_sched_barrier(0);  // (keeps the reads of the later columns from being hoisted)
      }
      __builtin_amdgcn_sched_barrier(0);
      mid();
      __builtin_amdgcn_sched_barrier(0);
      row_pass(zp, t_h == 0 ? 6 : 0);   // xi 1 (aa = 1) / xi 3 (aa = 0)
      __builtin_amdgcn_sched_barrier(0);
      row_pass(zm, t_h == 0 ? 12 : 6);  // xi 2 (aa = 2) / xi 4 (aa = 1)
    }
  };

  // ---- MFMA stage: wave = (channel half, xi half, nu half) owns positions (xi0 + i / 3, nu0 + i % 3), i = 0..8, x 32 channels ----
  const int m_nh = wave & 1, m_g = wave >> 1;
  const int m_pos0 = (3 * (m_g >> 1)) * 6 + 3 * (m_g & 1);
  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const size_t pos_stride = (size_t)a.nchunks * a.cout * CF_BK;
  const float* const wbase = a.weight + (size_t)m_pos0 * pos_stride + (size_t)(n0 / 32 + m_nh) * 512;  // (wave-uniform)
  f32x4 bq[4][2];  // ring of four positions: [slot][hi, lo]
  auto load_B = [&](int chunk, int i) __attribute__((always_inline)) {
    const float* wc = wbase + (size_t)((i / 3) * 6 + i % 3) * pos_stride + (size_t)chunk * a.cout * CF_BK + lane * 4;
    bq[i & 3][0] = *reinterpret_cast<const f32x4*>(wc);
    bq[i & 3][1] = *reinterpret_cast<const f32x4*>(wc + 256);
  };
  const float* const a_lane = V + m_pos0 * F4_PS + l31 * CF_LDK + half * 4;
  f32x4 va[4][2];  // A fragments of four positions: [slot][hi, lo] (row = tile l31, channels half*8 .. +7)
  auto read_A = [&](int i) __attribute__((always_inline)) {
    const float* p = a_lane + ((i / 3) * 6 + i % 3) * F4_PS;
    va[i & 3][0] = *reinterpret_cast<const f32x4*>(p);
    va[i & 3][1] = *reinterpret_cast<const f32x4*>(p + 8);
  };
  auto mfma = [&](f32x4 av, f32x4 bv, f32x16& c) __attribute__((always_inline)) {
#if F4_ABLATE & 1
    c[0] += av[0] + bv[1];
#else
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f4_f16x8, av), __builtin_bit_cast(f4_f16x8, bv), c, 0, 0, 0);
#endif
  };
  // lo*hi + hi*lo + hi*hi per position (the order of the F(2,3) kernels), two positions interleaved: no MFMA waits for its predecessor
  auto mma2 = [&](int i, int j) __attribute__((always_inline)) {
    mfma(va[i & 3][1], bq[i & 3][0], acc[i]);
    mfma(va[j & 3][1], bq[j & 3][0], acc[j]);
    mfma(va[i & 3][0], bq[i & 3][1], acc[i]);
    mfma(va[j & 3][0], bq[j & 3][1], acc[j]);
    mfma(va[i & 3][0], bq[i & 3][0], acc[i]);
    mfma(va[j & 3][0], bq[j & 3][0], acc[j]);
  };
  auto mma_stage = [&](int chunk) __attribute__((always_inline)) {  // B(0), B(1) of this slab were requested before the transform
#if !(F4_ABLATE & 16)
    load_B(chunk, 2);
    load_B(chunk, 3);
#endif
    read_A(0);
    read_A(1);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      read_A(2 * p + 2);
      if (2 * p + 3 < 9) read_A(2 * p + 3);
      __builtin_amdgcn_sched_barrier(0);
      mma2(2 * p, 2 * p + 1);
#if !(F4_ABLATE & 16)
      if (2 * p + 4 < 9) load_B(chunk, 2 * p + 4);  // refill of the slot just consumed
      if (2 * p + 5 < 9) load_B(chunk, 2 * p + 5);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma(va[0][1], bq[0][0], acc[8]);
    mfma(va[0][0], bq[0][1], acc[8]);
    mfma(va[0][0], bq[0][0], acc[8]);
  };

  // ---- slab loop ----
  load_A(0);
  __syncthreads();  // (the GroupNorm rows are in LDS)
#if !(F4_ABLATE & 4)
  store_patch(0);
#endif
  for (int s = 0; s < n; ++s) {
    load_A(s + 1 < n ? s + 1 : s);  // a whole slab ahead (clamped on the last one)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // patch(s) visible; every wave is done with V(s - 1)
#if !(F4_ABLATE & 2)
    transform([&]() __attribute__((always_inline)) {
#if !(F4_ABLATE & 16)
      load_B(s, 0);
      load_B(s, 1);
#endif
    });
#elif !(F4_ABLATE & 16)
    load_B(s, 0);
    load_B(s, 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // V(s) visible; the patch buffer is free
    mma_stage(s);
    __builtin_amdgcn_sched_barrier(0);
#if !(F4_ABLATE & 4)
    if (s + 1 < n) store_patch(s + 1);
#endif
  }

#if F4_ABLATE & 8
  if (a.sft_w != 12345.f) return;
#endif
  // ---- epilogue: four passes (channel half nh, tile half th) through LDS: M[36 positions][16 tiles][32 channels (+4)] ----
  float* const Mst = V;
  const int e_a = wave & 3;                                // output row of the 4x4 tile (wave-uniform)
  const int e_t16 = (wave >> 2) * 8 + (lane >> 3);         // tile within the pass's 16
  const int e_q = lane & 7;                                // channel quad within the pass's 32 channels
  const float acc_s = a.acc_scale * act_is;                // (a product of powers of two: exact)
  const unsigned e_rowc = (unsigned)a.w * (unsigned)a.cout;
#pragma unroll 1
  for (int nh = 0; nh < 2; ++nh) {
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const int nn = n0 + nh * 32 + e_q * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + nn);
#pragma unroll
    for (int th = 0; th < 2; ++th) {  // (unrolled: `th` selects accumulator registers)
      // residual / SFT operands of this pass first: their latency overlaps the staging
      const int tile = th * 16 + e_t16;
      const unsigned off0 = (((unsigned)b * a.h + (y0 + 4 * (tile >> 3) + e_a)) * a.w + (x0 + 4 * (tile & 7))) * (unsigned)a.cout + nn;
      f32x4 r0[4], r1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        r0[c] = r1[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.epilogue == CF_EPI_RESIDUAL || a.epilogue == CF_EPI_SFT) r0[c] = *reinterpret_cast<const f32x4*>(a.res + off0 + c * a.cout);
        if (a.epilogue == CF_EPI_SFT) r1[c] = *reinterpret_cast<const f32x4*>(a.sft_scale + off0 + c * a.cout);
      }
      __syncthreads();  // the previous pass's reads (first pass: the last MMA stage's reads of V) are complete
      if (m_nh == nh) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          float* mp = Mst + ((m_pos0 + (i / 3) * 6 + i % 3) * 16 + 4 * half) * F4_RLD + l31;
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) {
            const float val = th == 0 ? acc[i][r8] : acc[i][8 + r8];
            mp[((r8 & 3) + 8 * (r8 >> 2)) * F4_RLD] = val;
          }
        }
      }
      __syncthreads();
      // item (output row e_a, tile, channel quad): xi axis first (six nu columns), then the nu axis
      const float* mq = Mst + e_t16 * F4_RLD + e_q * 4;
      auto M4 = [&](int xi, int nu) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(mq + ((xi * 6 + nu) * 16) * F4_RLD); };
      f32x4 rr[6];
      if (e_a == 0) {
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) rr[nu] = ((M4(0, nu) + M4(1, nu)) + (M4(2, nu) + M4(3, nu))) + M4(4, nu);
      } else if (e_a == 1) {
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) rr[nu] = (M4(1, nu) - M4(2, nu)) * 0.5f + (M4(3, nu) - M4(4, nu)) * 2.f;
      } else if (e_a == 2) {
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) rr[nu] = (M4(1, nu) + M4(2, nu)) * 0.25f + (M4(3, nu) + M4(4, nu)) * 4.f;
      } else {
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) rr[nu] = ((M4(1, nu) - M4(2, nu)) * 0.125f + (M4(3, nu) - M4(4, nu)) * 8.f) + M4(5, nu);
      }
      const f32x4 s1 = rr[1] + rr[2], d1 = rr[1] - rr[2], s2 = rr[3] + rr[4], d2 = rr[3] - rr[4];
      f32x4 o[4];
      o[0] = (rr[0] + s1) + s2;
      o[1] = d1 * 0.5f + d2 * 2.f;
      o[2] = s1 * 0.25f + s2 * 4.f;
      o[3] = (d1 * 0.125f + d2 * 8.f) + rr[5];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 v = o[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * acc_s + bias4[e];
        if (a.epilogue == CF_EPI_RESIDUAL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r0[c][e];
        } else if (a.epilogue == CF_EPI_SFT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = r0[c][e] + a.sft_w * (r0[c][e] * r1[c][e] + v[e]);
        }
        *reinterpret_cast<f32x4*>(a.out + off0 + c * a.cout) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ssum[e] += v[e];
          ssq[e] += v[e] * v[e];
        }
      }
    }
    if (a.stats_out) {
      // GroupNorm statistics of the values this wave wrote for channel half nh (output row e_a of eight tile columns x two tile
      // halves x 2 passes): fp64 partials, fixed shuffle order; one partial per (image, group, patch, wave): nparts = tiles_per_img * 8
      const int cpg = a.stats_cpg;
      double d0, q0, d1v = 0, q1v = 0;
      if (cpg == 2) {
        d0 = (double)ssum[0] + ssum[1];
        q0 = (double)ssq[0] + ssq[1];
        d1v = (double)ssum[2] + ssum[3];
        q1v = (double)ssq[2] + ssq[3];
      } else {
        d0 = ((double)ssum[0] + ssum[1]) + ((double)ssum[2] + ssum[3]);
        q0 = ((double)ssq[0] + ssq[1]) + ((double)ssq[2] + ssq[3]);
      }
      for (int o2 = 8; o2 < 64; o2 <<= 1) {  // the eight tiles of this wave: lanes with the same channel quad
        d0 += __shfl_xor(d0, o2, 64);
        q0 += __shfl_xor(q0, o2, 64);
      }
      if (cpg == 2) {
        for (int o2 = 8; o2 < 64; o2 <<= 1) {
          d1v += __shfl_xor(d1v, o2, 64);
          q1v += __shfl_xor(q1v, o2, 64);
        }
      }
      for (int o2 = 1; o2 * 4 < cpg; o2 <<= 1) {  // adjacent channel quads of one group (cpg >= 8)
        d0 += __shfl_xor(d0, o2, 64);
        q0 += __shfl_xor(q0, o2, 64);
      }
      if ((lane >> 3) == 0 && (nn % cpg) == 0) {
        const size_t pidx = (size_t)rt * 8 + wave;
        const int ng = a.cout / cpg;
        double* op = a.stats_out + (((size_t)b * ng + nn / cpg) * a.nparts + pidx) * 2;
        op[0] = d0;
        op[1] = q0;
        if (cpg == 2) {
          op[(size_t)a.nparts * 2] = d1v;
          op[(size_t)a.nparts * 2 + 1] = q1v;
        }
      }
    }
  }
}

// U' = scale * G' g G'^T (fp64, rounded once to fp32) as hi = f16(U'), lo = f16(U' - hi), in MFMA-operand order
// [pos = xi*6 + nu][cin_pad/16][cout_pad/32][part: hi, lo][lane 64][4 words]; a lane's 16 bytes are the 8 halves of
// U'[n = tile*32 + (lane&31)][c = chunk*16 + (lane>>5)*8 + 0..7]  (v_mfma_f32_32x32x16_f16 B operand).
__global__ void pack_weight_wf43_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, int nchunks, float scale,
                                        unsigned* __restrict__ packed, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one 32-bit word = two halves
  if (i >= total) return;
  const int e = (int)(i & 3), ln = (int)((i >> 2) & 63), part = (int)((i >> 8) & 1);
  long r = i >> 9;
  const int ntiles = cout_pad / 32;
  const int nn = (int)(r % ntiles) * 32 + (ln & 31);
  r /= ntiles;
  const int chunk = (int)(r % nchunks);
  const int pos = (int)(r / nchunks);
  const int xi = pos / 6, nu = pos % 6;
  // rows of G' = D^-1 G for the points (0, 1/2, -1/2, 2, -2, inf), D = diag(1/4, 1/4, 1/4, 1/2, 1/2, 1/4)
  const double Gm[6][3] = {{4.0, 0.0, 0.0},           {-32.0 / 15.0, -16.0 / 15.0, -8.0 / 15.0}, {-32.0 / 15.0, 16.0 / 15.0, -8.0 / 15.0},
                           {1.0 / 15.0, 2.0 / 15.0, 4.0 / 15.0}, {1.0 / 15.0, -2.0 / 15.0, 4.0 / 15.0},  {0.0, 0.0, 4.0}};
  unsigned out = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = chunk * CF_BK + (ln >> 5) * 8 + e * 2 + h;
    float val = 0.f;
    if (nn < cout && c < cin) {
      const float* g = w + ((long)nn * cin + c) * 9;
      double u = 0.0;
#pragma unroll
      for (int y = 0; y < 3; ++y) {
        double rowv = 0.0;
#pragma unroll
        for (int x = 0; x < 3; ++x) rowv += (double)g[y * 3 + x] * Gm[nu][x];
        u += Gm[xi][y] * rowv;
      }
      val = (float)(u * (double)scale);
    }
    const _Float16 hi = (_Float16)val;
    const _Float16 hv = part ? (_Float16)(val - (float)hi) : hi;
    out |= (unsigned)__builtin_bit_cast(unsigned short, hv) << (16 * h);
  }
  packed[i] = out;
}

}  // namespace

extern "C" int cf_pack_conv_weight_winograd43_f16x2(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed,
                                                    cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_winograd43_f16x2: null pointer");
  CF_REQUIRE(cin_pad % CF_BK == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 64 == 0,
             "cf_pack_conv_weight_winograd43_f16x2: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  int ex = 0;
  CF_REQUIRE(scale > 0.f && frexpf(scale, &ex) == 0.5f, "cf_pack_conv_weight_winograd43_f16x2: scale %g is not a power of two", (double)scale);
  const long total = 36L * cin_pad * cout_pad;  // 32-bit words: hi + lo half per weight
  hipLaunchKernelGGL(pack_weight_wf43_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin,
                     cout_pad, cin_pad / CF_BK, scale, reinterpret_cast<unsigned*>(packed), total);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_winograd43_f16x2");
  return CF_OK;
}

// Called by cf_conv2d (cf_igemm.hip) for descriptors with winograd == 2; the common argument checks have run there.
int cf_wf43_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query) {
  CF_REQUIRE(d->taps == 9 && d->stride == 1 && !d->upsample && !d->in_nchw && !d->out_nchw && d->bf16_mfma == CF_OPERAND_F16X2,
             "cf_conv2d(winograd 2): F(4x4,3x3) covers 3x3 stride-1 NHWC convolutions with split-half operands");
  CF_REQUIRE(d->acc_scale > 0.f, "cf_conv2d(winograd 2): acc_scale must be the inverse of the pack-time weight scale (got %g)", (double)d->acc_scale);
  CF_REQUIRE(d->hout % F4_TH == 0 && d->wout % F4_TW == 0, "cf_conv2d(winograd 2): needs an output of %dx%d multiples (got %dx%d)", F4_TH,
             F4_TW, d->hout, d->wout);
  CF_REQUIRE(d->cout % F4_BN == 0 && d->cout_pad == d->cout, "cf_conv2d(winograd 2): needs cout == cout_pad, a multiple of 64 (got %d / %d)",
             d->cout, d->cout_pad);
  CF_REQUIRE(d->c0 + d->c1 <= F4_TAB, "cf_conv2d(winograd 2): at most %d input channels (got %d)", F4_TAB, d->c0 + d->c1);
  CF_REQUIRE(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_RESIDUAL || d->epilogue == CF_EPI_SFT,
             "cf_conv2d(winograd 2): epilogues are none / residual / SFT");
  CF_REQUIRE(d->pad_mode == CF_PAD_ZERO && (d->ld_in0 == 0 || d->ld_in0 == d->c0) && (d->ld_in1 == 0 || d->ld_in1 == d->c1) &&
                 (d->ld_out == 0 || d->ld_out == d->cout) && d->split_k < 1,
             "cf_conv2d(winograd 2): reads / writes dense tensors with zero padding, no split_k");
  CF_REQUIRE(d->stats_cpg == 0 || d->stats_cpg <= 32, "cf_conv2d(winograd 2): stats_cpg %d", d->stats_cpg);
  F4Args a;
  a.in0 = d->in0;
  a.in1 = d->in1;
  a.c0 = d->c0;
  a.c1 = d->c1;
  a.cin = d->c0 + d->c1;
  a.nchunks = a.cin / CF_BK;
  a.batch = d->batch;
  a.h = d->hout;
  a.w = d->wout;
  a.cout = d->cout;
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
  a.stats_cpg = d->stats_cpg > 0 ? d->stats_cpg : 2;
  a.tiles_x = d->wout / F4_TW;
  a.tiles_per_img = a.tiles_x * (d->hout / F4_TH);
  a.nparts = a.tiles_per_img * 8;
  a.ntn = d->cout / F4_BN;
  if (parts_query) {
    *parts_query = a.nparts;
    return CF_OK;
  }
  constexpr size_t lds = F4_LDS_FLOATS * sizeof(float);
  CF_LDS_ATTR((wf43_kernel<CF_PRO_NONE>), lds);  // (cf_device_init sets the dynamic-LDS attribute on each device)
  CF_LDS_ATTR((wf43_kernel<CF_PRO_AFFINE>), lds);
  CF_LDS_ATTR((wf43_kernel<CF_PRO_AFFINE_SWISH>), lds);
  CF_LDS_ATTR((wf43_kernel<CF_PRO_LEAKY>), lds);
  const dim3 grid(a.tiles_per_img * d->batch * a.ntn), block(F4_THREADS);
  switch (d->prologue) {
    case CF_PRO_AFFINE: hipLaunchKernelGGL((wf43_kernel<CF_PRO_AFFINE>), grid, block, lds, stream, a); break;
    case CF_PRO_AFFINE_SWISH: hipLaunchKernelGGL((wf43_kernel<CF_PRO_AFFINE_SWISH>), grid, block, lds, stream, a); break;
    case CF_PRO_LEAKY: hipLaunchKernelGGL((wf43_kernel<CF_PRO_LEAKY>), grid, block, lds, stream, a); break;
    default: hipLaunchKernelGGL((wf43_kernel<CF_PRO_NONE>), grid, block, lds, stream, a); break;
  }
  CF_CHECK_LAUNCH("cf_conv2d(winograd F(4,3) f16x2)");
  return CF_OK;
}
