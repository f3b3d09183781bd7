// Winograd F(2x2,3x3) convolution with split-half operands for the 64-OUTPUT-CHANNEL layers (64->64 and 128->64 at 512x512:
// vqgan_arch.py:243-262 encoder blocks 1-2, :300-316 generator blocks 21-22), as a PERSISTENT, DMA-fed streaming kernel (gfx950).
//
// These layers are HBM-bound by nature (64->64 @512^2 x16: 3.2 GB per launch against 0.16 ms of f16 MFMA work) and the four-wave
// form of cf_winograd.hip ran them at 0.30 of the HBM peak: a workgroup prefetches ONE 16-channel slab (11.5 KB) ahead, two
// workgroups per CU, so about 23 KB are in flight per CU and every slab exposes a loaded-HBM round trip (~2.4 us) -- a latency
// bound, not a bandwidth or an ALU bound.  A wave cannot prefetch deeper in registers, and it cannot keep deep LDS-DMA requests in
// flight either while it also waits for its weight fragments: vmcnt is ONE in-order counter.  So this kernel splits the roles:
//   * one LOADER wave per workgroup issues nothing but `global_load_lds_dwordx4` (LDS-DMA: no registers, 1 KB per instruction)
//     into an eight-slot ring of raw fp32 slabs and keeps FIVE slabs (60 KB per CU) in flight across barriers with counted
//     `s_waitcnt vmcnt(48)` -- its counter sees nothing else;
//   * eight COMPUTE waves (wave = (xi, 32-channel half), as in cf_wsplit.hip) never load activations from memory: per slab g they run
//        M(g): 12 MFMAs from V[g & 1] | P(g+2): GroupNorm / swish prologue IN PLACE on ring slot g+2 | T(g+1): slot -> V[(g+1) & 1]
//     between two barriers; their only global loads are the weight fragments of the next slab (L2) and the residual operands;
//   * the workgroup is persistent: it walks a contiguous run of output patches (strip order: four vertically adjacent patches, then
//     the next column, so both halo directions are re-read from L2), the ring and the loader run across patch boundaries, only the
//     compute side drains for the epilogue (its output-domain staging reuses the V buffers).
// V holds the transformed input ALREADY SPLIT into hi / lo IEEE halves, converted once by the transform (the eight-wave kernel of
// cf_wsplit.hip converts every A fragment in both channel-half waves) in MFMA-operand order [pos][k half][hi | lo][tile][8 halves]:
// a lane's fragment is two conflict-free ds_read_b128.  The ring slot layout is free under LDS-DMA (the destination is lane-linear,
// the SOURCE address is per lane): pixel (hy, hx) lives at 16-byte position 4 * ps + (quad ^ ((ps >> 2) & 3)), ps = 19 hy + hx +
// ((hy >> 1) & 1), which makes the transform's strided row reads conflict-free for the hardware's ds_read_b128 lane groups.
// Arithmetic (transform, split, MFMA order, output transform, epilogue, statistics) is that of winograd_kernel<false, true>:
// outputs and statistics partials are BITWISE those of the four-wave kernel (tools/split_ab.py AB_COMPARE=1).
#include <type_traits>

#include "cf_common.h"

#ifndef CF_W64
#define CF_W64 1   // 0: cf_w64_covers() answers false (A/B builds of tools/split_ab.sh)
#endif
// X_ABLATE: timing-only ablation builds (tools/split_ab.sh); 0 / undefined in every product build.
// 1 no MMA stage, 2 no prologue stage, 4 no transform, 8 no epilogue arithmetic / stores, 16 no DMA, 32 no weight fetch
#ifndef X_ABLATE
#define X_ABLATE 0
#endif

namespace {

constexpr int X_TH = 8, X_TW = 16;
constexpr int X_PITCH = 19;                       // px-slots per halo row (18 pixels + the parity shift)
constexpr int X_NT = 32;                          // Winograd tiles per patch
constexpr int X_SLOT_BYTES = 192 * 64;            // 192 px-slots x 16 channels fp32
constexpr int X_SLOT_FLOATS = X_SLOT_BYTES / 4;
constexpr int X_RING = 8;
constexpr int X_LEAD = 4;                         // slabs in flight (4 x 13 = 52 outstanding requests <= 63)
constexpr int X_CHUNKS = 12;                      // 1 KB DMA instructions per slab
constexpr int X_OPS = X_CHUNKS + 1;               // + one byte-wide DMA that pulls 64 cache lines of the patch's residual tile into L2
constexpr int X_V_BYTES = 32768;                  // 16 pos x 2 k halves x (hi | lo) x 32 tiles x 16 B
constexpr int X_V_FLOATS = X_V_BYTES / 4;
constexpr int X_CWAVES = 8;
constexpr int X_THREADS = 64 * (X_CWAVES + 1);    // 576
constexpr int X_LDS_BYTES = X_RING * X_SLOT_BYTES + 2 * X_V_BYTES;  // 163840 = all of the CU's LDS
static_assert(X_LDS_BYTES <= 160 * 1024, "ring + V buffers must fit the 160 KiB LDS");
static_assert(X_LEAD * X_OPS <= 63, "vmcnt counts at most 63 outstanding requests");

typedef _Float16 x_f16x8 __attribute__((ext_vector_type(8)));

struct W64Args {
  const float* in0;
  int cin, nchunks;
  int batch, h, w;
  int epilogue;
  const float* pro_scale;
  const float* pro_shift;
  const float* weight;  // [16 pos][nchunks][2 n tiles][hi, lo][64 lanes][4 words]  (cf_pack_conv_weight_winograd_f16x2, cout = 64)
  const float* bias;
  const float* res;
  float acc_scale;
  const float* act_scale;
  float* out;
  double* stats_out;
  int nparts;
  int tiles_x, tiles_per_img, strip_rows;
  int npatches, per_wg;
};

struct PatchPos {
  int b, y0, x0, rt;
};

// strip order: `strip_rows` vertically adjacent patches, then the next column of the strip; rt = raster index inside the image
__device__ __forceinline__ PatchPos x_decode(const W64Args& a, int q) {
  PatchPos p;
  p.b = q / a.tiles_per_img;
  const int r = q - p.b * a.tiles_per_img;
  const int per_strip = a.strip_rows * a.tiles_x;
  const int strip = r / per_strip;
  const int rem = r - strip * per_strip;
  const int col = rem / a.strip_rows;
  const int row = strip * a.strip_rows + (rem - col * a.strip_rows);
  p.y0 = row * X_TH;
  p.x0 = col * X_TW;
  p.rt = row * a.tiles_x + col;
  return p;
}

// 16-byte item i (0 .. 767) of a ring slot -> halo pixel (hy, hx) and channel quad; hx < 0: the position holds no pixel
__device__ __forceinline__ void x_slot_item(int i, int& hy, int& hx, int& quad) {
  const int ps = i >> 2;
  hy = ps / X_PITCH;
  hx = ps - hy * X_PITCH - ((hy >> 1) & 1);
  if (hy > 9 || hx > 17) hx = -1;
  quad = (i & 3) ^ ((ps >> 2) & 3);
}
// byte offset of (hy, hx, quad) inside a slot
__device__ __forceinline__ int x_slot_off(int hy, int hx, int quad) {
  const int ps = hy * X_PITCH + hx + ((hy >> 1) & 1);
  return (ps * 4 + (quad ^ ((ps >> 2) & 3))) * 16;
}

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + lane * 16 (M0 is written in the same statement)
__device__ __forceinline__ void x_glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);  // ("s" needs a provably wave-uniform value)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// 64 lanes x 1 byte: a prefetch -- each lane touches one cache line, the bytes land in 64 B of LDS nobody reads
__device__ __forceinline__ void x_glds1(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int PRO>
__global__ __launch_bounds__(X_THREADS) void w64_kernel(const W64Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const ring = smem;
  float* const Vb = smem + X_RING * X_SLOT_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.nchunks;

  const int q0 = blockIdx.x * a.per_wg;
  const int npw = min(a.per_wg, a.npatches - q0);   // (>= 1: the host sizes the grid so)
  const int G = npw * n;                             // slabs of this workgroup

  // =================================================== loader wave ===================================================
  if (wave == X_CWAVES) {
    const unsigned ring_lds = (unsigned)(uintptr_t)ring;
    // this lane's twelve items of a slab: patch-independent (hy, hx, quad), packed hy | hx << 4 | quad << 9 (hx = 31: no pixel)
    int item[X_CHUNKS];
#pragma unroll
    for (int c = 0; c < X_CHUNKS; ++c) {
      int hy, hx, quad;
      x_slot_item(c * 64 + lane, hy, hx, quad);
      item[c] = hy | ((hx < 0 ? 31 : hx) << 4) | (quad << 9);
    }
    unsigned goff[X_CHUNKS];  // element offsets of this lane's items for the patch being loaded (tensors < 16 GiB)
    // The slot's last 12 items hold no pixel (ps >= 189).  Items 756..759 / 760..763 carry the slab's 16 GroupNorm scale / shift values
    // instead (lanes 52..59 of the last DMA instruction read pro_scale / pro_shift): the prologue stage needs no global load at all.
    constexpr bool affine_l = PRO == CF_PRO_AFFINE || PRO == CF_PRO_AFFINE_SWISH;
    const bool tab_lane = affine_l && lane >= 52 && lane < 60;
    const float* tab_ptr = a.in0;
    // Residual prefetch: the output tile's residual operands (128 pixels x 256 B = 256 cache lines) are pulled into this XCD's L2 by
    // one byte-wide DMA per slab (64 lines each: slab s takes lines 64 (s & 3) ..), several slab steps before the epilogue asks for
    // them -- the compute waves cannot hold them in registers that long (168 VGPRs), and requested late they expose an HBM round trip.
    const bool has_res = a.epilogue == CF_EPI_RESIDUAL;
    unsigned roff = 0;  // element offset of this lane's line 0 .. 63 of the patch's residual tile (line = 2 * pixel + half)
    auto set_patch = [&](int q) __attribute__((always_inline)) {
      const PatchPos p = x_decode(a, q);
      roff = (((unsigned)p.b * a.h + p.y0 + (lane >> 5)) * a.w + p.x0 + ((lane >> 1) & 15)) * 64u + (lane & 1) * 32u;
      if (tab_lane) tab_ptr = (lane < 56 ? a.pro_scale : a.pro_shift) + (size_t)p.b * a.cin + (lane & 3) * 4;
      const unsigned center = (((unsigned)p.b * a.h + p.y0) * a.w + p.x0) * (unsigned)a.cin;
#pragma unroll
      for (int c = 0; c < X_CHUNKS; ++c) {
        const int hy = item[c] & 15, hx = (item[c] >> 4) & 31, quad = item[c] >> 9;
        const int iy = p.y0 - 1 + hy, ix = p.x0 - 1 + hx;
        const bool ok = hx != 31 && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
        goff[c] = ok ? (((unsigned)p.b * a.h + iy) * a.w + ix) * (unsigned)a.cin + quad * 4 : center;  // (zeroed by the prologue stage)
      }
    };
    int gl = 0, ql = q0, sl = 0;  // next slab to issue: stream index, patch, slab of the patch
    set_patch(ql);
    auto issue = [&]() __attribute__((always_inline)) {
      if (gl >= G) return;
#if X_ABLATE & 16
      ++gl;
      return;
#endif
      const unsigned dst = ring_lds + (unsigned)(gl & (X_RING - 1)) * X_SLOT_BYTES;
      const float* src = a.in0 + sl * CF_BK;
#pragma unroll
      for (int c = 0; c < X_CHUNKS - 1; ++c) x_glds16(src + goff[c], dst + c * 1024);
      x_glds16(tab_lane ? tab_ptr + sl * CF_BK : src + goff[X_CHUNKS - 1], dst + (X_CHUNKS - 1) * 1024);
      // lines 64 (sl & 3) + lane: two pixel rows further down per step
      x_glds1(has_res ? a.res + roff + (unsigned)(sl & 3) * 2u * (unsigned)a.w * 64u : a.in0, dst + 764 * 16);
      ++gl;
      if (++sl == n) {
        sl = 0;
        if (++ql < q0 + npw) set_patch(ql);
      }
    };
    // one step per barrier of the compute side that needs a new slab: the oldest outstanding slab has landed afterwards
    int done = 0;  // slabs known to have landed
    auto step = [&]() __attribute__((always_inline)) {
      const int outstanding = (X_ABLATE & 16) ? 0 : gl - done;
      static_assert(X_OPS == 13 && X_LEAD == 4, "the counted waits below assume 13 requests per slab, at most four slabs outstanding");
      if (outstanding >= 4) asm volatile("s_waitcnt vmcnt(39)" ::: "memory");
      else if (outstanding == 3) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
      else if (outstanding == 2) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (done < gl) ++done;
      __syncthreads();
      issue();  // (after the barrier: the compute waves must not wait for twelve DMA issues; the slot of slab gl - 8 has been free for a while)
    };
#pragma unroll 1
    for (int i = 0; i < X_LEAD; ++i) issue();
    step();  // S1: slab 0
    step();  // S2: slab 1
    step();  // S3: slab 2
#pragma unroll 1
    for (int pi = 0; pi < npw; ++pi) {
#pragma unroll 1
      for (int k = 0; k < n; ++k) {
        __syncthreads();  // first half-step of slab g
        step();           // barrier closing slab g: slab g + 3 has landed
      }
      __syncthreads();                     // E1
      __syncthreads();                     // E2
      __syncthreads();                     // E3
    }
    return;
  }

  // =================================================== compute waves ==================================================
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int xi = wave & 3;
  const int nh = wave >> 2;       // 32-channel half of this wave (waves w and w + 4 share a SIMD)
  const int gtid = tid & 255;

  // ---- P stage: items tid and tid + 512 (the latter for tid < 256) of a slot, in place.  Thread-derived geometry is recomputed
  // from the lane where it is used (per patch / per slab): kept in registers across the slab loop it was spilled, and a scratch
  // reload inside the loop is a vmcnt(0) behind the weight refill ----
  const bool p_second = wave < 4;
  constexpr bool affine = PRO == CF_PRO_AFFINE || PRO == CF_PRO_AFFINE_SWISH;
  int gp = 0, qp = q0, sp = 0;   // P cursor: next slab to activate (stream index, patch, slab of the patch)
  bool p_valid[2];
  float act_s = 1.f, act_s02 = 0.2f;
  auto p_set_patch = [&](int q) __attribute__((always_inline)) {
    const PatchPos p = x_decode(a, q);
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int hy, hx, quad;
      x_slot_item(wave * 64 + ln + 512 * j, hy, hx, quad);
      const int iy = p.y0 - 1 + hy, ix = p.x0 - 1 + hx;
      p_valid[j] = hx >= 0 && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
    }
    if (!affine && a.act_scale) {
      act_s = a.act_scale[2 * p.b];
      act_s02 = 0.2f * act_s;
    }
  };
  auto p_stage = [&]() __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    float* slot = ring + (gp & (X_RING - 1)) * X_SLOT_FLOATS;
    float* item = slot + (wave * 64 + ln) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && !p_second) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(item + 2048 * j);
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (affine) {  // the slab's GroupNorm rows travel in the slot's tail (written by the loader's DMA): broadcast LDS reads
        // channel quad of item i: (i & 3) ^ ((i >> 4) & 3)  (x_slot_item); i = 64 wave + lane + 512 j: the j term leaves both fields alone
        const int quad = (ln & 3) ^ (((wave * 64 + ln) >> 4) & 3);
        sc = *reinterpret_cast<const f32x4*>(slot + (756 + quad) * 4);
        sh = *reinterpret_cast<const f32x4*>(slot + (760 + quad) * 4);
      }
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
        v[e] = p_valid[j] ? y : 0.f;
      }
      // (items 756 .. 767 of a slot hold no pixel: 756 .. 763 are the GroupNorm rows the other threads are reading)
      if (j == 0 || wave * 64 + ln + 512 < 756) *reinterpret_cast<f32x4*>(item + 2048 * j) = v;
    }
    ++gp;
    if (++sp == n) {
      sp = 0;
      if (++qp < q0 + npw) p_set_patch(qp);
    }
  };

  // ---- T stage: item (tile, channel quad c4, xi row) per thread; the lane -> tile map follows the ds_read_b128 lane groups ----
  int t_tile;
  {
    const int l32 = lane & 31;
    int grp, j;
    if (l32 < 4) { grp = 0; j = l32; }
    else if (l32 < 12) { grp = 1; j = l32 - 4; }
    else if (l32 < 16) { grp = 0; j = l32 - 8; }
    else if (l32 < 20) { grp = 1; j = l32 - 8; }
    else if (l32 < 28) { grp = 0; j = l32 - 12; }
    else { grp = 1; j = l32 - 16; }
    t_tile = grp * 16 + ((((j >> 3) ^ grp) & 1) << 3) + (j & 7);
  }
  const int t_c4 = (wave & 1) * 2 + (lane >> 5);
  const int t_xi = wave >> 1;
  // B^T d along rows:  xi0 = r0 - r2, xi1 = r1 + r2, xi2 = r2 - r1, xi3 = r1 - r3, each as q + s*p (one FMA per value, exact)
  const int t_q = t_xi == 0 ? 0 : (t_xi == 2 ? 2 : 1);
  const int t_p = t_xi == 2 ? 1 : (t_xi == 3 ? 3 : 2);
  const float t_s = t_xi == 1 ? 1.f : -1.f;
  int t_offq[4], t_offp[4];  // byte offsets in a slot of the item's 2 rows x 4 columns
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    t_offq[c] = x_slot_off(2 * (t_tile >> 3) + t_q, 2 * (t_tile & 7) + c, t_c4);
    t_offp[c] = x_slot_off(2 * (t_tile >> 3) + t_p, 2 * (t_tile & 7) + c, t_c4);
  }
  const int t_offv = (t_xi * 4) * 2048 + (t_c4 >> 1) * 1024 + t_tile * 16 + (t_c4 & 1) * 8;  // byte offset in V of (pos (xi, 0), hi)
  auto t_stage = [&](int g) __attribute__((always_inline)) {
    const char* slot = reinterpret_cast<const char*>(ring + (g & (X_RING - 1)) * X_SLOT_FLOATS);
    char* V = reinterpret_cast<char*>(Vb + (g & 1) * X_V_FLOATS) + t_offv;
    f32x4 t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 dq = *reinterpret_cast<const f32x4*>(slot + t_offq[c]);
      const f32x4 dp = *reinterpret_cast<const f32x4*>(slot + t_offp[c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) t[c][e] = __fmaf_rn(dp[e], t_s, dq[e]);
      if (c == 1) __builtin_amdgcn_sched_barrier(0);  // two column pairs: all eight reads at once cost 32 registers at the loop's pressure peak
    }
    // (.) B along columns: nu0 = t0 - t2, nu1 = t1 + t2, nu2 = t2 - t1, nu3 = t1 - t3; each value split ONCE into hi + lo halves
    const f32x4 v[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      float h0, l0, h1, l1;
      cf_split_pair(v[nu][0], v[nu][1], h0, l0);
      cf_split_pair(v[nu][2], v[nu][3], h1, l1);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f32x2*>(V + nu * 2048) = f32x2{h0, h1};
      *reinterpret_cast<f32x2*>(V + nu * 2048 + 512) = f32x2{l0, l1};
    }
  };

  // ---- M stage: wave (xi, nh) owns positions (xi, 0..3) x 32 channels ----
  f32x16 acc[4];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
  // weight fragments: wave-uniform base (SGPR pair) + one 32-bit lane offset (the saddr form of global_load): per-position 64-bit
  // VGPR pointers were hoisted out of the slab loop and spilled
  const size_t pos_stride = (size_t)a.nchunks * 64 * CF_BK;
  const float* const wwave = a.weight + (size_t)(xi * 4) * pos_stride + (size_t)nh * 512;
  const unsigned wlane_off = lane * 16;
  f32x4 bq[4][2];  // [nu][hi, lo]
  auto load_B = [&](int chunk, int nu) __attribute__((always_inline)) {
    // (readfirstlane pins the uniform pointer to an SGPR pair: otherwise the loop's strength reduction keeps four 64-bit VGPR pointers)
    const unsigned long long u = reinterpret_cast<unsigned long long>(wwave + (size_t)chunk * 64 * CF_BK + nu * pos_stride);
    const char* wc = reinterpret_cast<const char*>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32) |
                                                   __builtin_amdgcn_readfirstlane((unsigned)u));
    bq[nu][0] = *reinterpret_cast<const f32x4*>(wc + wlane_off);
    bq[nu][1] = *reinterpret_cast<const f32x4*>(wc + 1024 + wlane_off);
  };
  const int a_off = (xi * 4) * 2048 + half * 1024 + l31 * 16;  // byte offset of this lane's hi fragment of position (xi, 0)
  f32x4 va[2][2];
  auto read_A = [&](const char* V, int nu) __attribute__((always_inline)) {
    va[nu & 1][0] = *reinterpret_cast<const f32x4*>(V + a_off + nu * 2048);        // hi: 8 halves
    va[nu & 1][1] = *reinterpret_cast<const f32x4*>(V + a_off + nu * 2048 + 512);  // lo
  };
  auto m_stage = [&](int g, int next_chunk) __attribute__((always_inline)) {
    const char* V = reinterpret_cast<const char*>(Vb + (g & 1) * X_V_FLOATS);
    read_A(V, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      if (nu < 3) read_A(V, nu + 1);
      __builtin_amdgcn_sched_barrier(0);
      const x_f16x8 ah = __builtin_bit_cast(x_f16x8, va[nu & 1][0]), al = __builtin_bit_cast(x_f16x8, va[nu & 1][1]);
      acc[nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(x_f16x8, bq[nu][0]), acc[nu], 0, 0, 0);
      acc[nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(x_f16x8, bq[nu][1]), acc[nu], 0, 0, 0);
      acc[nu] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(x_f16x8, bq[nu][0]), acc[nu], 0, 0, 0);
#if !(X_ABLATE & 32)
      load_B(next_chunk, nu);  // refill for the next slab (the next patch starts at slab 0 again): a whole iteration of cover
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue geometry (the four-wave kernel's, one 32-channel pass per wave group) ----
  float* const R = Vb + nh * (8 * X_NT * 32);  // [(xi*2 + bb)][tile][32 channels], channels rotated by 16 for bb = 1 (bank spread)
  const int n0 = nh * 32;

  // ---- start-up ----
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) load_B(0, nu);
  p_set_patch(qp);
  __syncthreads();  // S1: slab 0 has landed
  p_stage();        // P(0)
  __syncthreads();  // S2: slab 1
  if (G > 1) p_stage();  // P(1)
  t_stage(0);
  __syncthreads();  // S3: slab 2

  int g = 0;
#pragma unroll 1
  for (int pi = 0; pi < npw; ++pi) {
    const PatchPos pm = x_decode(a, q0 + pi);
    // One slab = TWO half-steps with a barrier each.  The two waves that share a SIMD (w and w + 4: the two channel halves) alternate:
    // in a half-step one of them runs its MMA stage M(k) while the other runs its prologue / transform items P(k + 2), T(k + 1), then they
    // swap -- the matrix pipe and the vector ALU / LDS of every SIMD are busy at the same time, with ONE copy of either stage's code
    // (separate straight-line copies per role cost ~100 spilled registers at this kernel's 168).  Dependencies: T(k) of both groups is
    // complete two half-steps before the first M(k); V[(k + 1) & 1] is rewritten only after both groups' M(k - 1).
#pragma unroll 1
    for (int h = 0; h < 2 * n; ++h) {
      const int k = h >> 1;
      if (((h ^ nh) & 1) == 0) {
#if !(X_ABLATE & 1)
        m_stage(g + k, k + 1 < n ? k + 1 : 0);
#endif
      } else {
#if !(X_ABLATE & 2)
        if (g + k + 2 < G) p_stage();
#endif
        __builtin_amdgcn_sched_barrier(0);
#if !(X_ABLATE & 4)
        if (k + 1 < n) t_stage(g + k + 1);
#endif
      }
      __syncthreads();
    }
    g += n;
    // output offsets: one laundered base per thread (see below), the (k2, aa) variants differ by wave-uniform amounts
    unsigned offs[4];
    f32x4 r0[4];
    {
      int gt = lane;
      asm volatile("" : "+v"(gt));
      gt += xi * 64;
      const unsigned e_rowc = (unsigned)a.w * 64u;
      const unsigned e_base = (((unsigned)pm.b * a.h + (pm.y0 + 2 * ((gt >> 4) >> 3))) * a.w + (pm.x0 + 2 * ((gt >> 4) & 7) + ((gt >> 3) & 1))) * 64u +
                              (n0 + (gt & 7) * 4);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) offs[k2 * 2 + aa] = e_base + (unsigned)(k2 * 4 + aa) * e_rowc;
    }
    // residual operands: L2 hits by now (the loader's byte-wide DMAs), requested ahead of the staging exchange that covers them
    if (a.epilogue == CF_EPI_RESIDUAL) {
#pragma unroll
      for (int i = 0; i < 4; ++i) r0[i] = *reinterpret_cast<const f32x4*>(a.res + offs[i]);
    }
    // ---- to the output domain: nu axis in registers, xi axis through LDS (the V buffers are idle now).  The staging addresses are
    // one laundered base per thread plus compile-time offsets (DS immediate fields): left to itself hipcc hoists thirty-odd
    // thread-derived addresses of this (inner) epilogue above the patch loop and spills them ----
#if X_ABLATE & 8
    __syncthreads();
    __syncthreads();
    __syncthreads();
    if (a.acc_scale != 12345.f) continue;
#endif
    int ln = lane;
    asm volatile("" : "+v"(ln));  // (everything below derives from the laundered lane: nothing can be hoisted above the patch loop)
    const int e_l31 = ln & 31, e_half = ln >> 5, e_gtid = xi * 64 + ln, e_n4 = ln & 7, e_bb = (ln >> 3) & 1;
    const int wb0 = ((xi * 2 + 0) * X_NT + 4 * e_half) * 32 + e_l31;          // R[xi][0][row = 4 half + ..][channel]
    const int wb1 = ((xi * 2 + 1) * X_NT + 4 * e_half) * 32 + (e_l31 ^ 16);   // R[xi][1]: channels rotated by 16 (bank spread of the reads)
    const int rb = (e_bb * X_NT + (e_gtid >> 4)) * 32 + ((e_n4 ^ (e_bb * 4)) * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
      const int roff = ((r & 3) + 8 * (r >> 2)) * 32;  // (cf_acc_row without its lane term)
      R[wb0 + roff] = (m0 + m1) + m2;  // R[xi][0] = M0 + M1 + M2
      R[wb1 + roff] = (m1 - m2) - m3;  // R[xi][1] = M1 - M2 - M3
    }
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][r] = 0.f;
    __syncthreads();  // E1
    f32x4 o[4];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {  // item (tile = gtid / 16 + 16 k2, output column bb, channel quad): two output rows
      f32x4 x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(R + rb + (q * 2 * X_NT + 16 * k2) * 32);
      o[k2 * 2 + 0] = (x[0] + x[1]) + x[2];  // xi axis: Y[0][bb] = R0 + R1 + R2 ; Y[1][bb] = R1 - R2 - R3
      o[k2 * 2 + 1] = (x[1] - x[2]) - x[3];
    }
    __syncthreads();  // E2: staging reads are done, the next patch's first transform may rewrite V
    if (pi + 1 < npw) t_stage(g);
    // ---- bias / residual, 16-byte stores, GroupNorm statistics of what was written (cpg = 2: a lane's four channels are two groups) ----
    {
      const int nn = n0 + e_n4 * 4;
      f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + nn);
      float act_is = 1.f;
      if (!affine && a.act_scale) act_is = a.act_scale[2 * pm.b + 1];
      const float acc_s = a.acc_scale * act_is;  // (a product of powers of two: exact)
      float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = o[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * acc_s + bias4[e];
        if (a.epilogue == CF_EPI_RESIDUAL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r0[i][e];
        }
        *reinterpret_cast<f32x4*>(a.out + offs[i]) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ssum[e] += v[e];
          ssq[e] = __fmaf_rn(v[e], v[e], ssq[e]);  // (what hipcc contracts the four-wave kernel's  ssq += v * v  into: same bits)
        }
      }
      if (a.stats_out) {
        double d0 = (double)ssum[0] + ssum[1], q0s = (double)ssq[0] + ssq[1];
        double d1 = (double)ssum[2] + ssum[3], q1s = (double)ssq[2] + ssq[3];
        for (int o2 = 8; o2 < 64; o2 <<= 1) {  // the (tile, bb) items of this wave: lanes with the same channel quad
          d0 += __shfl_xor(d0, o2, 64);
          q0s += __shfl_xor(q0s, o2, 64);
        }
        for (int o2 = 8; o2 < 64; o2 <<= 1) {
          d1 += __shfl_xor(d1, o2, 64);
          q1s += __shfl_xor(q1s, o2, 64);
        }
        if ((lane >> 3) == 0) {
          const size_t pidx = (size_t)pm.rt * 4 + xi;
          double* op = a.stats_out + (((size_t)pm.b * 32 + nn / 2) * a.nparts + pidx) * 2;
          op[0] = d0;
          op[1] = q0s;
          op[(size_t)a.nparts * 2] = d1;
          op[(size_t)a.nparts * 2 + 1] = q1s;
        }
      }
    }
    __syncthreads();  // E3: V[g & 1] of the next patch is visible
  }
}

}  // namespace

// Called by cf_winograd_launch (cf_winograd.hip) for split-half Winograd descriptors; the common argument checks have run there.
bool cf_w64_covers(const cf_conv_desc* d) {
#if CF_W64
  return d->winograd && d->bf16_mfma == CF_OPERAND_F16X2 && d->split_k < 1 && d->cout == 64 && d->cout_pad == 64 && d->c1 == 0 &&
         d->c0 % CF_BK == 0 && d->hout % X_TH == 0 && d->wout % X_TW == 0 && (long)d->hout * d->wout >= 128 * 128 &&
         (d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_RESIDUAL) && (d->stats_cpg == 0 || d->stats_cpg == 2);
#else
  (void)d;
  return false;
#endif
}

int cf_w64_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query) {
  W64Args a;
  a.in0 = d->in0;
  a.cin = d->c0;
  a.nchunks = d->c0 / CF_BK;
  a.batch = d->batch;
  a.h = d->hout;
  a.w = d->wout;
  a.epilogue = d->epilogue;
  a.pro_scale = d->pro_scale;
  a.pro_shift = d->pro_shift;
  a.weight = d->weight;
  a.bias = d->bias;
  a.res = d->res;
  a.acc_scale = d->acc_scale;
  a.act_scale = d->act_scale;
  a.out = d->out;
  a.stats_out = d->stats_out;
  a.tiles_x = d->wout / X_TW;
  const int tiles_y = d->hout / X_TH;
  a.tiles_per_img = a.tiles_x * tiles_y;
  a.nparts = a.tiles_per_img * 4;
  if (parts_query) {
    *parts_query = a.nparts;
    return CF_OK;
  }
  a.strip_rows = tiles_y % 4 == 0 ? 4 : (tiles_y % 2 == 0 ? 2 : 1);
  a.npatches = a.tiles_per_img * d->batch;
  const int cus = cf_device_cu_count() > 0 ? cf_device_cu_count() : 256;
  a.per_wg = (a.npatches + cus - 1) / cus;
  const int grid = (a.npatches + a.per_wg - 1) / a.per_wg;
  static unsigned long long attr_devs = 0;  // bit d: the LDS attribute has been set on device d
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 64 || !((attr_devs >> dev) & 1ull)) {
    hipError_t e = hipSuccess;
    const void* const kerns[4] = {reinterpret_cast<const void*>(w64_kernel<CF_PRO_NONE>), reinterpret_cast<const void*>(w64_kernel<CF_PRO_AFFINE>),
                                  reinterpret_cast<const void*>(w64_kernel<CF_PRO_AFFINE_SWISH>), reinterpret_cast<const void*>(w64_kernel<CF_PRO_LEAKY>)};
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES);
    if (e != hipSuccess) {
      cf_set_error("cf_conv2d(winograd f16x2, 64 channels): hipFuncSetAttribute(%d B LDS): %s", X_LDS_BYTES, hipGetErrorString(e));
      return CF_ERR_LAUNCH;
    }
    if (dev < 64) attr_devs |= 1ull << dev;
  }
  const dim3 g(grid), block(X_THREADS);
  switch (d->prologue) {
    case CF_PRO_AFFINE: hipLaunchKernelGGL((w64_kernel<CF_PRO_AFFINE>), g, block, X_LDS_BYTES, stream, a); break;
    case CF_PRO_AFFINE_SWISH: hipLaunchKernelGGL((w64_kernel<CF_PRO_AFFINE_SWISH>), g, block, X_LDS_BYTES, stream, a); break;
    case CF_PRO_LEAKY: hipLaunchKernelGGL((w64_kernel<CF_PRO_LEAKY>), g, block, X_LDS_BYTES, stream, a); break;
    default: hipLaunchKernelGGL((w64_kernel<CF_PRO_NONE>), g, block, X_LDS_BYTES, stream, a); break;
  }
  CF_CHECK_LAUNCH("cf_conv2d(winograd f16x2, 64 channels)");
  return CF_OK;
}
