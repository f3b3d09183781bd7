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
// Work decomposition (512 threads = 8 waves, one workgroup per CU, 157 KB LDS):
//   * a workgroup owns a 16x32 output patch of ONE image (4x8 tiles of 4x4 outputs) x 64 channels; K loop over 16-channel slabs;
//   * two wave groups, A = waves 0..3 and B = waves 4..7, each own ONE xi half of the transform domain (18 positions): a group
//     transforms its half of V and runs the MFMAs on it, so a V half never crosses groups -- and the groups run half a slab out of
//     phase:       phase 1:  A: transform(s)                     | B: MMA(s - 1), prologue + store(s + 1)
//                  phase 2:  A: MMA(s), prologue + store(s + 1)  | B: transform(s)            (one barrier after each phase)
//     so on every SIMD (it hosts one wave of each group) the LDS-bound transform of one wave runs beside the MFMA / weight-fragment /
//     swish work of the other.  The stage-synchronous first version of this kernel (all waves: transform | MMA, store) measured
//     1.17 ms on 64 -> 64 @ 512x512 x 16 with the stages adding up: loads + epilogue alone 0.51 ms (6.3 TB/s), MFMAs 0.12, transform
//     0.19, prologue + store 0.25, weight fetch 0.15 (profiles/r04_f43_sync_ablation.txt);
//   * gather: the 18x34 halo patch of a slab (612 pixels x 4 channel quads, five float4 items per thread) is requested a whole slab
//     ahead (registers), passes the GroupNorm-apply / swish or LeakyReLU prologue, zero padding and concat as in the other kernels, and
//     is written to one of TWO patch buffers (slab parity): [18 rows x 36 pixel slots][16 floats], unpadded; the 64-byte pixel slot p
//     lives at p ^ ((p >> 2) & 3) (low two bits), which spreads the four tile columns a half-wave reads over the four 64-byte windows
//     of the 256-byte bank row;
//   * input transform: item = (tile, channel pair) of the group's xi half: column pass for three rows of B'^T d (five of the six tile
//     rows are read), row pass for their six nu, split into hi + lo and written to V[36][32 tiles][16 hi halves | 16 lo halves]
//     (64-byte rows, 16-byte chunk c of tile t at c ^ ((t >> 2) & 3): conflict-free for the lane groups of ds_read_b128);
//   * MFMA stage: wave = (xi half = group, nu half, channel half): nine positions x 32 channels = 144 accumulator registers; A
//     fragments from V, B fragments global/L2 -> registers through a four-position ring, MFMAs of two positions interleaved;
//   * epilogue: the accumulators go through LDS in two passes (16 tiles each, [36][16 tiles][64 ch], over patch buffers + V); item =
//     (tile, channel pair) reads its 36 transform-domain values ONCE, contracts xi and nu, applies acc_scale / bias / residual / SFT,
//     stores the 4x4 pixels as 8-byte pairs (a half-wave writes 256 contiguous bytes per pixel) and accumulates the GroupNorm
//     statistics of what it wrote (fp32 over four values, then fp64; fixed shuffle order: eight partials per patch and group).
#include <type_traits>

#include "cf_common.h"

#ifndef F4_ABLATE   // timing-only ablation builds: 1 no MFMAs, 2 no transform, 4 no prologue + store, 8 no epilogue, 16 no weight fetch
#define F4_ABLATE 0
#endif

namespace {

constexpr int F4_TH = 16, F4_TW = 32;            // output patch of a workgroup
constexpr int F4_PW = F4_TW + 2;                 // halo patch 18 x 34
constexpr int F4_NPIX = (F4_TH + 2) * F4_PW;     // 612
constexpr int F4_PWL = 36;                       // pixel slots per patch row in LDS (a multiple of 4: the swizzle stays inside a row)
constexpr int F4_SLOTS = (F4_TH + 2) * F4_PWL;   // 648
constexpr int F4_DUMMY = 646;                    // an unused slot (row 17 holds pixels in slots 612..645): target of the padding items
constexpr int F4_NT = 32;                        // tiles per patch (4 rows x 8 columns) = one MFMA row tile
constexpr int F4_THREADS = 512;
constexpr int F4_BN = 64;                        // output channels per workgroup
constexpr int F4_APT = 5;                        // float4 gather items per thread: 640 items x 4 quads / 512 threads
constexpr int F4_PATCH_FLOATS = F4_SLOTS * CF_BK;   // 10368 floats = 41472 bytes per buffer
constexpr int F4_PS = F4_NT * CF_BK;             // 512 floats between positions of V
constexpr int F4_V_FLOATS = 36 * F4_PS;          // 18432
constexpr int F4_M_FLOATS = 36 * 16 * F4_BN;     // 36864: one pass of the epilogue (36 positions x 16 tiles x 64 channels)
constexpr int F4_TAB = 512;                      // GroupNorm scale / shift rows of the image (cin <= 512)
constexpr int F4_LDS_FLOATS = 2 * F4_PATCH_FLOATS + F4_V_FLOATS + 2 * F4_TAB;   // 160,768 bytes
static_assert(F4_M_FLOATS <= 2 * F4_PATCH_FLOATS + F4_V_FLOATS, "epilogue staging must fit the patch buffers + V");
static_assert(F4_LDS_FLOATS * 4 <= 163840, "LDS budget");

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
  float* const V = smem + 2 * F4_PATCH_FLOATS;
  float* const tab = V + F4_V_FLOATS;  // [scale: F4_TAB][shift: F4_TAB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform by construction: keeps what depends on it in SGPRs)
  const int grp = wave >> 2;                                   // wave group = xi half
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
  // One word per item: bits 0..9 the (swizzled) LDS pixel slot, bits 10..30 the pixel's index inside the image, bit 31 = outside the
  // image or a padding item (p >= 612): loaded from pixel 0, stored as zeros (padding items: into an unused slot).
  const int k4 = tid & 3;
  unsigned pcode[F4_APT];
#pragma unroll
  for (int j = 0; j < F4_APT; ++j) {
    const int p = (tid >> 2) + 128 * j;
    unsigned code = 0x80000000u;
    int slot = F4_DUMMY;
    if (p < F4_NPIX) {
      const int hy = p / F4_PW;
      const int hx = p - hy * F4_PW;
      const int iy = y0 - 1 + hy;
      const int ix = x0 - 1 + hx;
      slot = hy * F4_PWL + hx;
      if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) code = (unsigned)(iy * a.w + ix) << 10;
    }
    pcode[j] = code | (unsigned)((slot & ~3) | ((slot ^ (slot >> 2)) & 3));
  }
  const size_t img0 = (size_t)b * a.h * a.w;
  f32x4 ra[F4_APT];
  // unconditional loads from clamped addresses; out-of-image items are zeroed at the store (see cf_winograd.hip)
  auto load_A = [&](int chunk) __attribute__((always_inline)) {
    const int c = chunk * CF_BK + k4 * 4;
    const bool first = c < a.c0;
    const int cs = first ? a.c0 : a.c1;
    const float* src = (first ? a.in0 : a.in1) + img0 * cs + (first ? c : c - a.c0);
#pragma unroll
    for (int j = 0; j < F4_APT; ++j) ra[j] = *reinterpret_cast<const f32x4*>(src + (size_t)((pcode[j] & 0x7fffffffu) >> 10) * cs);
  };
  auto store_patch = [&](int chunk) __attribute__((always_inline)) {
    float* const pb = smem + (chunk & 1) * F4_PATCH_FLOATS + k4 * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
      sc = *reinterpret_cast<const f32x4*>(tab + chunk * CF_BK + k4 * 4);
      sh = *reinterpret_cast<const f32x4*>(tab + F4_TAB + chunk * CF_BK + k4 * 4);
    }
#pragma unroll
    for (int j = 0; j < F4_APT; ++j) {
      const bool valid = (int)pcode[j] >= 0;
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
      *reinterpret_cast<f32x4*>(pb + (pcode[j] & 1023u) * CF_BK) = v;
    }
  };

  // ---- input transform of this group's xi half: item = (tile (ty, tx), channel pair cp): V[(3 grp + aa) * 6 + nu], aa = 0..2, nu = 0..5 ----
  const int t_tx = lane >> 3, t_cp = lane & 7, t_ty = wave & 3;
  // pixel (tile row r, tile column j) sits in slot (4 ty + r) * 36 + 4 tx + j, stored at its low two bits ^ ((slot >> 2) & 3) =
  // ^ ((r + tx) & 3) for j < 4 and ^ ((r + tx + 1) & 3) for j = 4, 5 (36 ty * 9 and 9 r reduce to r mod 4): byte offset of column j
  // inside the row = (j << 6) ^ sw[r & 3], resp. 256 + (((j - 4) << 6) ^ sw[(r + 1) & 3]).
  const unsigned t_base = (unsigned)(((4 * t_ty) * F4_PWL + 4 * t_tx) * 64 + t_cp * 8);  // bytes from the patch buffer: tile row 0, column 0
  unsigned sw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sw[k] = (unsigned)((t_tx + k) & 3) << 6;
  const int t_tile = t_ty * 8 + t_tx;
  const int t_st = (t_tile >> 2) & 3;
  // hi word of channel pair cp: chunk cp >> 2 (0, 1), lo word: chunk 2 + (cp >> 2); chunks swizzled by the tile
  float* const t_hi = V + (grp * 18) * F4_PS + t_tile * CF_BK + (((t_cp >> 2) ^ t_st) << 2) + (t_cp & 3);
  float* const t_lo = V + (grp * 18) * F4_PS + t_tile * CF_BK + (((2 + (t_cp >> 2)) ^ t_st) << 2) + (t_cp & 3);
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
      t_hi[(pos + nu) * F4_PS] = hi;
      t_lo[(pos + nu) * F4_PS] = lo;
    }
  };
  // Two parts keep the live set small (the accumulators hold 144 of the 256 registers): first the single row of the half (xi 0 from
  // tile rows 0, 2, 4 / xi 5 from rows 1, 3, 5), then the even / odd pair (xi 1, 2 or 3, 4: tile rows 1..4).  `mid` runs between the
  // column and the row pass of the second part (the slab's first weight fragments are requested there: fewest live registers).
  auto transform = [&](int chunk, auto mid) __attribute__((always_inline)) {
    const char* const pb = reinterpret_cast<const char*>(smem + (chunk & 1) * F4_PATCH_FLOATS) + t_base;
    auto px = [&](int r, int j) __attribute__((always_inline)) {  // (r, j compile-time after unrolling)
      const unsigned off = j < 4 ? (((unsigned)j << 6) ^ sw[r & 3]) : (256u + (((unsigned)(j - 4) << 6) ^ sw[(r + 1) & 3]));
      return *reinterpret_cast<const f4_f32x2*>(pb + r * (F4_PWL * 64) + off);
    };
    if (grp == 0) {
      {  // xi = 0: .25 (d0 + d4) - 1.0625 d2
        f4_f32x2 z[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) z[j] = (px(0, j) + px(4, j)) * 0.25f - px(2, j) * 1.0625f;
        __builtin_amdgcn_sched_barrier(0);
        row_pass(z, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      {  // xi 1, 2: e = .25 d4 - d2, o = .125 d3 - .5 d1
        f4_f32x2 zp[6], zm[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f4_f32x2 e = px(4, j) * 0.25f - px(2, j);
          const f4_f32x2 o = px(3, j) * 0.125f - px(1, j) * 0.5f;
          zp[j] = e + o;
          zm[j] = e - o;
          if (j == 2) __builtin_amdgcn_sched_barrier(0);  // (keeps the reads of the later columns from being hoisted)
        }
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zp, 6);
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zm, 12);
      }
    } else {
      {  // xi = 5: .25 (d1 + d5) - 1.0625 d3   (position 30 = 18 + 12)
        f4_f32x2 z[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) z[j] = (px(1, j) + px(5, j)) * 0.25f - px(3, j) * 1.0625f;
        __builtin_amdgcn_sched_barrier(0);
        row_pass(z, 12);
      }
      __builtin_amdgcn_sched_barrier(0);
      {  // xi 3, 4: e = .5 d4 - .125 d2, o = d3 - .25 d1   (positions 18, 24)
        f4_f32x2 zp[6], zm[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f4_f32x2 e = px(4, j) * 0.5f - px(2, j) * 0.125f;
          const f4_f32x2 o = px(3, j) - px(1, j) * 0.25f;
          zp[j] = e + o;
          zm[j] = e - o;
          if (j == 2) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zp, 0);
        __builtin_amdgcn_sched_barrier(0);
        row_pass(zm, 6);
      }
    }
  };

  // ---- MFMA stage: wave = (xi half = grp, nu half, channel half) owns positions (3 grp + i / 3, 3 nuh + i % 3), i = 0..8, x 32 channels ----
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
#if !(F4_ABLATE & 16)
    const float* wc = wbase + (size_t)((i / 3) * 6 + i % 3) * pos_stride + (size_t)chunk * a.cout * CF_BK + lane * 4;
    bq[i & 3][0] = *reinterpret_cast<const f32x4*>(wc);
    bq[i & 3][1] = *reinterpret_cast<const f32x4*>(wc + 256);
#endif
  };
  const int m_st = (l31 >> 2) & 3;
  const float* const a_hi = V + m_pos0 * F4_PS + l31 * CF_BK + ((half ^ m_st) << 2);        // row = tile l31, channels half*8 .. +7
  const float* const a_lo = V + m_pos0 * F4_PS + l31 * CF_BK + (((2 + half) ^ m_st) << 2);
  f32x4 va[2][2];  // A fragments of two positions: [slot][hi, lo]
  auto read_A = [&](int i) __attribute__((always_inline)) {
    va[i & 1][0] = *reinterpret_cast<const f32x4*>(a_hi + ((i / 3) * 6 + i % 3) * F4_PS);
    va[i & 1][1] = *reinterpret_cast<const f32x4*>(a_lo + ((i / 3) * 6 + i % 3) * F4_PS);
  };
  auto mfma = [&](f32x4 av, f32x4 bv, f32x16& c) __attribute__((always_inline)) {
#if F4_ABLATE & 1
    c[0] += av[0] + bv[1];
#else
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f4_f16x8, av), __builtin_bit_cast(f4_f16x8, bv), c, 0, 0, 0);
#endif
  };
  // Position by position: lo*hi + hi*lo + hi*hi (the order of the F(2,3) kernels) into one accumulator -- back-to-back MFMAs with the
  // same destination forward their result; only one wave per SIMD is in this stage at a time, the other one fills the issue slots
  // with its transform.  A fragments one position ahead (two slots), B fragments four ahead (ring of four).
  auto mma_stage = [&](int chunk) __attribute__((always_inline)) {  // B(0), B(1) of this slab were requested inside the transform
    load_B(chunk, 2);
    load_B(chunk, 3);
    read_A(0);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (i + 1 < 9) read_A(i + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma(va[i & 1][1], bq[i & 3][0], acc[i]);
      mfma(va[i & 1][0], bq[i & 3][1], acc[i]);
      mfma(va[i & 1][0], bq[i & 3][0], acc[i]);
      if (i + 4 < 9) load_B(chunk, i + 4);  // refill of the slot just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto do_transform = [&](int s) __attribute__((always_inline)) {
#if !(F4_ABLATE & 2)
    transform(s, [&]() __attribute__((always_inline)) {
      load_B(s, 0);
      load_B(s, 1);
    });
#else
    load_B(s, 0);
    load_B(s, 1);
#endif
  };
  auto feed = [&](int s) __attribute__((always_inline)) {  // prologue + store of slab s (if any), then the request for slab s + 1
    if (s < n) {
#if !(F4_ABLATE & 4)
      store_patch(s);
#endif
      if (s + 1 < n) load_A(s + 1);
    }
  };

  // ---- slab loop: the two groups half a slab apart; every wave passes 2 n + 2 barriers ----
  load_A(0);
  __syncthreads();  // (the GroupNorm rows are in LDS)
  feed(0);
  __syncthreads();  // patch(0) visible
  if (grp == 0) {
    for (int s = 0; s < n; ++s) {
      do_transform(s);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // this group's V(s) visible; B is done with patch(s - 1)
      mma_stage(s);
      __builtin_amdgcn_sched_barrier(0);
      feed(s + 1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // patch(s + 1) complete (B stored its items a phase earlier)
    }
  } else {
    for (int s = 0; s < n; ++s) {
      if (s > 0) mma_stage(s - 1);
      __builtin_amdgcn_sched_barrier(0);
      feed(s + 1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // A is done with patch(s - 1) ... and this group's items of patch(s + 1) are written
      do_transform(s);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // this group's V(s) visible
    }
    mma_stage(n - 1);
  }

#if F4_ABLATE & 8
  if (a.sft_w != 12345.f) return;
#endif
  // ---- epilogue: two passes (tile half th) through LDS: M[36 positions][16 tiles][64 channels] over the patch buffers + V ----
  float* const Mst = smem;
  const int e_t16 = wave * 2 + half;                       // tile within the pass's 16
  const int e_cp = l31;                                    // channel pair within the 64 channels
  const float acc_s = a.acc_scale * act_is;                // (a product of powers of two: exact)
  const int nn = n0 + 2 * e_cp;
  f4_f32x2 bias2 = {0.f, 0.f};
  if (a.bias) bias2 = *reinterpret_cast<const f4_f32x2*>(a.bias + nn);
  const unsigned e_rowc = (unsigned)a.w * (unsigned)a.cout;
  double dsum = 0.0, dsq = 0.0, dsum1 = 0.0, dsq1 = 0.0;  // (channel 0 / 1 of the pair; joined below unless the group is one channel wide)
#pragma unroll
  for (int th = 0; th < 2; ++th) {  // (unrolled: `th` selects accumulator registers)
    const int tile = th * 16 + e_t16;
    const unsigned off0 = (((unsigned)b * a.h + (y0 + 4 * (tile >> 3))) * a.w + (x0 + 4 * (tile & 7))) * (unsigned)a.cout + nn;
    // residual / SFT operands of this pass first: their latency overlaps the staging
    f4_f32x2 r0[4][4], r1[4][4];
#pragma unroll
    for (int aa = 0; aa < 4; ++aa)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        r0[aa][c] = r1[aa][c] = f4_f32x2{0.f, 0.f};
        if (a.epilogue == CF_EPI_RESIDUAL || a.epilogue == CF_EPI_SFT)
          r0[aa][c] = *reinterpret_cast<const f4_f32x2*>(a.res + off0 + aa * e_rowc + c * a.cout);
        if (a.epilogue == CF_EPI_SFT) r1[aa][c] = *reinterpret_cast<const f4_f32x2*>(a.sft_scale + off0 + aa * e_rowc + c * a.cout);
      }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // the previous pass's reads (first pass: the last MMA stage's reads of V) are complete
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      float* mp = Mst + ((m_pos0 + (i / 3) * 6 + i % 3) * 16 + 4 * half) * F4_BN + m_nh * 32 + l31;
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) mp[((r8 & 3) + 8 * (r8 >> 2)) * F4_BN] = acc[i][th * 8 + r8];
    }
    __syncthreads();
    // item (tile, channel pair): the xi axis per nu column (t[aa] = sum_xi A^T[aa][xi] M[xi][nu]), columns folded into the 4x4 outputs
    const float* mq = Mst + e_t16 * F4_BN + 2 * e_cp;
    f4_f32x2 o[4][4];
    auto col = [&](int nu, f4_f32x2 (&t)[4]) __attribute__((always_inline)) {
      f4_f32x2 m[6];
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) m[xi] = *reinterpret_cast<const f4_f32x2*>(mq + ((xi * 6 + nu) * 16) * F4_BN);
      const f4_f32x2 s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
      t[0] = (m[0] + s1) + s2;
      t[1] = d1 * 0.5f + d2 * 2.f;
      t[2] = s1 * 0.25f + s2 * 4.f;
      t[3] = (d1 * 0.125f + d2 * 8.f) + m[5];
    };
    {
      f4_f32x2 t0[4], ta[4], tb[4];
      col(0, t0);
      col(1, ta);
      col(2, tb);
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) {
        const f4_f32x2 s = ta[aa] + tb[aa], d = ta[aa] - tb[aa];
        o[aa][0] = t0[aa] + s;
        o[aa][1] = d * 0.5f;
        o[aa][2] = s * 0.25f;
        o[aa][3] = d * 0.125f;
      }
      col(3, ta);
      col(4, tb);
      col(5, t0);
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) {
        const f4_f32x2 s = ta[aa] + tb[aa], d = ta[aa] - tb[aa];
        o[aa][0] += s;
        o[aa][1] += d * 2.f;
        o[aa][2] += s * 4.f;
        o[aa][3] += d * 8.f + t0[aa];
      }
    }
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
      f4_f32x2 rs = {0.f, 0.f}, rq = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f4_f32x2 v = o[aa][c] * acc_s + bias2;
        if (a.epilogue == CF_EPI_RESIDUAL) v += r0[aa][c];
        else if (a.epilogue == CF_EPI_SFT) v = r0[aa][c] + a.sft_w * (r0[aa][c] * r1[aa][c] + v);
        *reinterpret_cast<f4_f32x2*>(a.out + off0 + aa * e_rowc + c * a.cout) = v;
        rs += v;
        rq += v * v;
      }
      dsum += (double)rs[0];
      dsq += (double)rq[0];
      dsum1 += (double)rs[1];
      dsq1 += (double)rq[1];
    }
  }
  if (a.stats_out) {
    // GroupNorm statistics of the values this wave wrote (2 tiles x 2 passes x 64 channels): fp64 partials, fixed shuffle order; one
    // partial per (image, group, patch, wave): nparts = tiles_per_img * 8
    const int cpg = a.stats_cpg;
    dsum += dsum1;  // (the pair belongs to one group: cpg is even)
    dsq += dsq1;
    dsum += __shfl_xor(dsum, 32, 64);  // the wave's two tiles
    dsq += __shfl_xor(dsq, 32, 64);
    for (int o2 = 1; o2 * 2 < cpg; o2 <<= 1) {  // adjacent channel pairs of one group (cpg >= 4)
      dsum += __shfl_xor(dsum, o2, 64);
      dsq += __shfl_xor(dsq, o2, 64);
    }
    if (half == 0 && (nn % cpg) == 0) {
      const size_t pidx = (size_t)rt * 8 + wave;
      const int ng = a.cout / cpg;
      double* op = a.stats_out + (((size_t)b * ng + nn / cpg) * a.nparts + pidx) * 2;
      op[0] = dsum;
      op[1] = dsq;
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
  CF_REQUIRE((long)d->hout * d->wout <= (1L << 21), "cf_conv2d(winograd 2): at most 2^21 pixels per image (got %dx%d)", d->hout, d->wout);
  CF_REQUIRE(d->c0 + d->c1 <= F4_TAB, "cf_conv2d(winograd 2): at most %d input channels (got %d)", F4_TAB, d->c0 + d->c1);
  CF_REQUIRE(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_RESIDUAL || d->epilogue == CF_EPI_SFT,
             "cf_conv2d(winograd 2): epilogues are none / residual / SFT");
  CF_REQUIRE(d->pad_mode == CF_PAD_ZERO && (d->ld_in0 == 0 || d->ld_in0 == d->c0) && (d->ld_in1 == 0 || d->ld_in1 == d->c1) &&
                 (d->ld_out == 0 || d->ld_out == d->cout) && d->split_k < 1,
             "cf_conv2d(winograd 2): reads / writes dense tensors with zero padding, no split_k");
  CF_REQUIRE(d->stats_cpg == 0 || (d->stats_cpg <= 64 && d->stats_cpg % 2 == 0), "cf_conv2d(winograd 2): stats_cpg %d (even, at most 64)", d->stats_cpg);
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
