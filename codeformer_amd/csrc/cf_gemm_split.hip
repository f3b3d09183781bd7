// Linear / 1x1 GEMM on token matrices with split-half operands (fp32 = hi + lo IEEE halves, three f16 MFMAs per product, fp32
// accumulation -- the scheme of cf_split.hip) for gfx950:   out[M][N] = epilogue(acc_scale * A[M][K] . W'[N][K]^T + bias).
//
// Serves the Transformer's Linear layers (codeformer_arch.py:104-106,126,132,183,192: feat_emb, the q|k / v / out projections, the
// MLP, the logits head) when CodeFormer.gemm_precision = 'f16x2'.  The fp32-MFMA GEMM they otherwise run on (cf_igemm.hip, 64x64
// split-K tiles) reaches 58-75 TFLOP/s on these shapes; the matrix work here is 3/16 of it.
//
//   * workgroup = 4 waves (2 x 2), 64 x 64 output tile, wave tile 32 x 32 (one MFMA tile: 16 accumulator registers);
//   * NO LDS in the main loop: a lane loads its A fragment (row = token, 8 consecutive channels, 32 bytes) and its B fragments (the
//     weights are packed in MFMA-operand order, hi and lo halves pre-split and scaled by a power of two: 1 KB contiguous per wave
//     and fragment) straight from global memory / L2, four 16-wide k steps ahead in registers (two groups of four steps, double
//     buffered); A is split into halves in registers (cf_split_pair);
//   * K is always cut into virtual chunks of 128 values summed from zero and added in chunk order (cf_common.h, the contract of the
//     fp32 split-K GEMM): the bits do not depend on how many workgroups (1, 2, 4, 8) share a tile, so the host picks the split
//     count from the tiles in flight and results stay bitwise batch-invariant;
//   * epilogue: bias, exact-erf GELU or residual, 128-byte row segments per store instruction.
#include <cstdlib>
#include <type_traits>

#include "cf_common.h"

namespace {

typedef _Float16 gs_f16x8 __attribute__((ext_vector_type(8)));

struct GsArgs {
  const float* a;     // [M][K] dense tokens
  const float* w;     // packed: [K/16][N/32][hi, lo][64 lanes][4 words]
  const float* bias;  // [N] or null
  const float* res;   // [M][N] (CF_EPI_RESIDUAL)
  float* out;         // [M][N]
  int M, N, K;
  int epilogue;
  float acc_scale;
  float* ws;
  unsigned* counters;
  int nsplit;
  // Two token matrices in one launch (ABI v22, cf_conv_desc.in0_alt): output columns >= n_alt are contracted against the rows of a_alt
  // instead of a -- the q|k projections read LayerNorm(x) + pos, the v projection LayerNorm(x) (codeformer_arch.py:125-126), one GEMM of
  // N = 3 E.  A tile never straddles n_alt (a multiple of 128); per output element nothing changes.  a_alt == nullptr: one matrix.
  const float* a_alt;
  int n_alt;
};
__device__ __forceinline__ const float* gs_rows(const GsArgs& g, int n0) { return (g.a_alt && n0 >= g.n_alt) ? g.a_alt : g.a; }

constexpr int GS_GROUP = 4;  // k steps (of 16) per prefetch group; a virtual chunk of 128 values = two groups

__global__ __launch_bounds__(256) void gemm_split_kernel(const GsArgs g) {
  __shared__ float s_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x;
  const int split = bid % g.nsplit;
  const int tile = bid / g.nsplit;
  const int ntn = g.N >> 6;
  const int nt = tile % ntn, mt = tile / ntn;
  const int m0 = mt * 64 + wm * 32, n0 = nt * 64 + wn * 32;
  const int V = g.K >> 7;            // virtual chunks of 128 K values
  const int nv = V / g.nsplit;       // ... of this workgroup (host-checked: nsplit divides V)
  const int c0 = split * nv;

  const float* const arow = gs_rows(g, n0) + (size_t)(m0 + l31) * g.K + half * 8 + (size_t)c0 * 128;
  const size_t kstride = (size_t)(g.N >> 5) * 512;  // floats between consecutive k steps of the packed weights
  const float* const wl = g.w + (size_t)(n0 >> 5) * 512 + lane * 4 + (size_t)c0 * 8 * kstride;

  f32x4 ra[2][GS_GROUP][2], rb[2][GS_GROUP][2];  // [buffer][k step][A: channels 0-3 / 4-7 of the lane's 8; B: hi / lo]
  auto fetch = [&](int grp, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int s = 0; s < GS_GROUP; ++s) {
      const int ks = grp * GS_GROUP + s;
      ra[BUF][s][0] = *reinterpret_cast<const f32x4*>(arow + ks * 16);
      ra[BUF][s][1] = *reinterpret_cast<const f32x4*>(arow + ks * 16 + 4);
      rb[BUF][s][0] = *reinterpret_cast<const f32x4*>(wl + (size_t)ks * kstride);
      rb[BUF][s][1] = *reinterpret_cast<const f32x4*>(wl + (size_t)ks * kstride + 256);
    }
  };
  f32x16 acc, tot;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = tot[r] = 0.f;
  auto compute = [&](auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int s = 0; s < GS_GROUP; ++s) {
      f32x4 ah, al;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 v = ra[BUF][s][e >> 1];
        float hh, ll;
        cf_split_pair(v[(e & 1) * 2], v[(e & 1) * 2 + 1], hh, ll);
        ah[e] = hh;
        al[e] = ll;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, al), __builtin_bit_cast(gs_f16x8, rb[BUF][s][0]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, ah), __builtin_bit_cast(gs_f16x8, rb[BUF][s][1]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, ah), __builtin_bit_cast(gs_f16x8, rb[BUF][s][0]), acc, 0, 0, 0);
    }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  const int ngroups = nv * 2;  // two groups per virtual chunk
  fetch(0, B0{});
  for (int c = 0; c < nv; ++c) {
    // first half of the chunk from buffer 0 while buffer 1 loads its second half; then the next chunk's first half into buffer 0
    fetch(2 * c + 1, B1{});
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc otherwise sinks every load to its use)
    compute(B0{});
    if (2 * c + 2 < ngroups) fetch(2 * c + 2, B0{});
    __builtin_amdgcn_sched_barrier(0);
    compute(B1{});
    // the chunk sum is complete: fold it into the running sum (one workgroup per tile) or park it for the last arriver
    if (g.nsplit == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[r] += acc[r];
    } else {
      f32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = f32x4{acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
      cf_splitk_park(v, g.ws, tile, c0 + c, V, 256);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  }
  if (g.nsplit > 1) {
    f32x4 v[4];
    if (!cf_splitk_finish(v, g.ws, g.counters, tile, V, g.nsplit, 256, &s_flag)) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) tot[4 * i + e] = v[i][e];
  }

  // ---- epilogue: lane holds column n0 + l31 of rows cf_acc_row(r, lane) ----
  const int n = n0 + l31;
  const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const size_t o = (size_t)(m0 + cf_acc_row(r, lane)) * g.N + n;
    float v = tot[r] * g.acc_scale + bias;
    if (g.epilogue == CF_EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (g.epilogue == CF_EPI_RESIDUAL) v += g.res[o];
    g.out[o] = v;
  }
}

// ---- the same GEMM with the token tile staged through LDS (round 4): 128 tokens x BN (64 / 128) columns per workgroup ------------
// The kernel above gathers its A fragments per lane (32 rows x 64 bytes per instruction) and keeps ONE MFMA tile per wave: at 4096 tokens
// it runs at 50-60 TFLOP/s, bound by the L2 latency of its 32-64 dependent k steps.  Here
//   * the 128 x 32 token tile of a stage is read coalesced (a row's 128 bytes by 8 threads), split ONCE into hi / lo halves and written
//     to LDS in operand order ([k step][row][half 0 hi | half 1 hi | half 0 lo | half 1 lo], 16-byte chunk c at c ^ ((row >> 2) & 3):
//     conflict-free ds_read_b128), double-buffered, one barrier per stage;
//   * a wave owns 64 tokens x BN / 2 columns (2 x NI MFMA tiles): a B fragment feeds two MFMA triples, an A fragment NI;
//   * B fragments straight from L2 in the packed operand order, one stage (two k steps) ahead in registers.
// Per output element the arithmetic is that of gemm_split_kernel with nsplit = 1 -- virtual chunks of 128 K values summed from zero in
// k order (lo*hi, hi*lo, hi*hi per step), chunk sums added in chunk order, the same epilogue expression -- so the two kernels agree
// BITWISE and the host picks by shape (this one: M % 128 == 0, no split) without touching batch invariance.
template <int NI>
__global__ __launch_bounds__(256) void gemm_split_tile_kernel(const GsArgs g) {
  constexpr int BN = 64 * NI;
  __shared__ __attribute__((aligned(16))) float As[2][2][128][16];  // [buffer][k step][row][16 words]: 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ntn = g.N / BN;
  const int nt = blockIdx.x % ntn, mt = blockIdx.x / ntn;
  const int m0 = mt * 128, n0 = nt * BN + wn * (32 * NI);
  const int nst = g.K >> 5;  // stages of 32 K values

  // staging: item j of this thread is float4 #q of row (tid + 256 j) >> 3 of the tile's stage (k = 4 q .. 4 q + 3)
  const float* const asrc = gs_rows(g, n0) + (size_t)(m0 + (tid >> 3)) * g.K + (tid & 7) * 4;
  // Operands in flight: THREE stages ahead in a ring of four register sets (a stage's MFMAs last ~0.2 us, a load 1-2 us: with one
  // stage ahead every stage waited for its operands -- 29.6 us on 4096 x 512 x 512, no better than the untiled kernel)
  f32x4 rg[4][4];
  auto load_stage = [&](int st, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) rg[BUF][j] = *reinterpret_cast<const f32x4*>(asrc + (size_t)(32 * j) * g.K + st * 32);
  };
  typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
  auto store_stage = [&](int lbuf, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
    const int q = tid & 7, ks = q >> 2, hf = (q >> 1) & 1, sub = q & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (tid >> 3) + 32 * j;
      const int swz = (row >> 2) & 3;
      float h0, l0, h1, l1;
      cf_split_pair(rg[BUF][j][0], rg[BUF][j][1], h0, l0);
      cf_split_pair(rg[BUF][j][2], rg[BUF][j][3], h1, l1);
      float* base = &As[lbuf][ks][row][0];
      *reinterpret_cast<gs_f32x2*>(base + ((hf ^ swz) << 2) + sub * 2) = gs_f32x2{h0, h1};
      *reinterpret_cast<gs_f32x2*>(base + (((2 + hf) ^ swz) << 2) + sub * 2) = gs_f32x2{l0, l1};
    }
  };
  const size_t kstride = (size_t)(g.N >> 5) * 512;  // floats between consecutive k steps of the packed weights
  const float* const wl = g.w + (size_t)(n0 >> 5) * 512 + lane * 4;
  f32x4 rb[4][2][NI][2];  // [ring slot][k step][n tile][hi, lo]
  auto load_B = [&](int st, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float* p = wl + (size_t)(st * 2 + s) * kstride + ni * 512;
        rb[BUF][s][ni][0] = *reinterpret_cast<const f32x4*>(p);
        rb[BUF][s][ni][1] = *reinterpret_cast<const f32x4*>(p + 256);
      }
  };
  f32x16 acc[2][NI], tot[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = tot[mi][ni][r] = 0.f;
  auto compute = [&](int lbuf, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 ah[2], al[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = wm * 64 + mi * 32 + l31;
        const int swz = (row >> 2) & 3;
        ah[mi] = *reinterpret_cast<const f32x4*>(&As[lbuf][s][row][(half ^ swz) << 2]);
        al[mi] = *reinterpret_cast<const f32x4*>(&As[lbuf][s][row][((2 + half) ^ swz) << 2]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, al[mi]), __builtin_bit_cast(gs_f16x8, rb[BUF][s][ni][0]), acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, ah[mi]), __builtin_bit_cast(gs_f16x8, rb[BUF][s][ni][1]), acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, ah[mi]), __builtin_bit_cast(gs_f16x8, rb[BUF][s][ni][0]), acc[mi][ni], 0, 0, 0);
        }
    }
  };
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>;
  using R3 = std::integral_constant<int, 3>;
  // stage st: ring slot `cur` holds its B fragments, LDS buffer st & 1 its tokens; slot `nxt` (stage st + 1) is split into the other
  // LDS buffer (its readers passed the barrier that opened this stage); the slot freed by stage st - 1 (`far`) takes stage st + 3
  // (whether a stage requests / splits operands is a compile-time flag: under a run-time condition inside the loop hipcc's merged wait
  //  counts make every stage wait for the loads it has just issued)
  auto stage = [&](int st, auto cur, auto nxt, auto far, auto do_load, auto do_store) __attribute__((always_inline)) {
    if constexpr (decltype(do_load)::value) {
      load_stage(st + 3, far);
      load_B(st + 3, far);
    }
    __builtin_amdgcn_sched_barrier(0);  // (keep the prefetch ahead of the MFMAs)
    compute(st & 1, cur);
    if ((st & 3) == 3) {  // a virtual chunk of 128 K values is complete: fold it into the running sum, restart from zero
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            tot[mi][ni][r] += acc[mi][ni][r];
            acc[mi][ni][r] = 0.f;
          }
    }
    if constexpr (decltype(do_store)::value) store_stage((st + 1) & 1, nxt);
    __syncthreads();
  };
  constexpr std::true_type Y{};
  constexpr std::false_type NO{};
  load_stage(0, R0{});
  load_B(0, R0{});
  load_stage(1, R1{});
  load_B(1, R1{});
  load_stage(2, R2{});
  load_B(2, R2{});
  store_stage(0, R0{});
  __syncthreads();
  int st = 0;
  for (; st + 4 < nst; st += 4) {  // (K % 128 == 0: four stages per virtual chunk)
    stage(st, R0{}, R1{}, R3{}, Y, Y);
    stage(st + 1, R1{}, R2{}, R0{}, Y, Y);
    stage(st + 2, R2{}, R3{}, R1{}, Y, Y);
    stage(st + 3, R3{}, R0{}, R2{}, Y, Y);
  }
  stage(st, R0{}, R1{}, R3{}, Y, Y);   // the last chunk: one stage left to request, three to split
  stage(st + 1, R1{}, R2{}, R0{}, NO, Y);
  stage(st + 2, R2{}, R3{}, R1{}, NO, Y);
  stage(st + 3, R3{}, R0{}, R2{}, NO, NO);

  // ---- epilogue: lane holds column n of rows cf_acc_row(r, lane) of each of its tiles (the expression of gemm_split_kernel) ----
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + ni * 32 + l31;
    const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const size_t o = (size_t)(m0 + wm * 64 + mi * 32 + cf_acc_row(r, lane)) * g.N + n;
        float v = tot[mi][ni][r] * g.acc_scale + bias;
        if (g.epilogue == CF_EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (g.epilogue == CF_EPI_RESIDUAL) v += g.res[o];
        g.out[o] = v;
      }
  }
}

// ---- the same GEMM with the K chunks of an output tile shared by the four waves of ONE workgroup (round 5): split_k == CF_SPLITK_IN_WORKGROUP (-1).
// The cross-workgroup split above costs a launch two dependent global phases (park + ticket, then the last arriver's ordered re-read: 16 us
// for 256 x 512 x 512), and the token-tile kernel walks its sixteen 32-wide stages one L2 round trip after the other (27 us for 4096 x 512
// x 512 at one wave per SIMD).  Here wave w accumulates virtual chunks w, w + 4, ... of a (32 MI) x (32 NI) tile from zero -- the operand
// fragments of a whole chunk (MI = NI = 1) or of half a chunk (2 x 2) requested at once: one or two L2 round trips per chunk --, parks each
// chunk sum in LDS, and after one barrier the 256 threads add the chunk sums IN CHUNK ORDER from zero and run the epilogue expression of the
// kernels above.  Per output element that is their arithmetic exactly (k steps in order inside a chunk, lo*hi, hi*lo, hi*hi per step; out =
// ((0 + P0) + P1) + ...), so all the kernels of this file agree BITWISE and the host may choose by the number of tiles in flight -- i.e.
// by the batch -- without touching batch invariance.  No workspace, no counters.
//   <1, 1>: 32 x 32 tiles -- the shipped instantiation, for launches of at most 2^20 outputs (one to four faces; at eight the N = 512 layers).
constexpr int GS_CHUNK_MAXV = 8;   // K <= 1024: chunk sums of a tile in LDS = V x MI x NI x 4 KB
template <int MI, int NI>
__global__ __launch_bounds__(256, MI * NI == 1 ? 1 : 2) void gemm_split_chunk_kernel(const GsArgs g) {
  extern __shared__ __attribute__((aligned(16))) float gs_part[];   // [chunk][mi][ni][accumulator register][lane]
  constexpr int HS = MI * NI == 1 ? 8 : 4;                          // k steps whose operands are in flight together
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ntn = g.N / (32 * NI);
  const int nt = blockIdx.x % ntn, mt = blockIdx.x / ntn;
  const int m0 = mt * (32 * MI), n0 = nt * (32 * NI);
  const int V = g.K >> 7;
  const float* const arow = gs_rows(g, n0) + (size_t)(m0 + l31) * g.K + half * 8;
  const size_t kstride = (size_t)(g.N >> 5) * 512;  // floats between consecutive k steps of the packed weights
  const float* const wl = g.w + (size_t)(n0 >> 5) * 512 + lane * 4;
  for (int c = wave; c < V; c += 4) {
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
    for (int h = 0; h < 8 / HS; ++h) {
      f32x4 ra[HS][MI][2], rb[HS][NI][2];
#pragma unroll
      for (int s = 0; s < HS; ++s) {
        const int ks = c * 8 + h * HS + s;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          ra[s][mi][0] = *reinterpret_cast<const f32x4*>(arow + (size_t)(32 * mi) * g.K + ks * 16);
          ra[s][mi][1] = *reinterpret_cast<const f32x4*>(arow + (size_t)(32 * mi) * g.K + ks * 16 + 4);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          rb[s][ni][0] = *reinterpret_cast<const f32x4*>(wl + (size_t)ks * kstride + ni * 512);
          rb[s][ni][1] = *reinterpret_cast<const f32x4*>(wl + (size_t)ks * kstride + ni * 512 + 256);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // every request of the group above the first MFMA (hipcc otherwise sinks each load next to its use)
#pragma unroll
      for (int s = 0; s < HS; ++s) {
        f32x4 ah[MI], al[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 v = ra[s][mi][e >> 1];
            float hh, ll;
            cf_split_pair(v[(e & 1) * 2], v[(e & 1) * 2 + 1], hh, ll);
            ah[mi][e] = hh;
            al[mi][e] = ll;
          }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, al[mi]), __builtin_bit_cast(gs_f16x8, rb[s][ni][0]), acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, ah[mi]), __builtin_bit_cast(gs_f16x8, rb[s][ni][1]), acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, ah[mi]), __builtin_bit_cast(gs_f16x8, rb[s][ni][0]), acc[mi][ni], 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) gs_part[(((c * MI + mi) * NI + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
  }
  __syncthreads();
  // thread t finishes accumulator registers (t >> 6) + 4 j of lane t & 63 of every tile: column n0 + 32 ni + (lane & 31), rows cf_acc_row(r, lane)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + ni * 32 + l31;
    const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wave + 4 * j;
        float tot = 0.f;
        for (int c = 0; c < V; ++c) tot += gs_part[(((c * MI + mi) * NI + ni) * 16 + r) * 64 + lane];
        const size_t o = (size_t)(m0 + mi * 32 + cf_acc_row(r, lane)) * g.N + n;
        float v = tot * g.acc_scale + bias;
        if (g.epilogue == CF_EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (g.epilogue == CF_EPI_RESIDUAL) v += g.res[o];
        g.out[o] = v;
      }
  }
}

// ---- the token-tile GEMM with IEEE-fp32 operands (round 6; precision 'fp32': the Transformer's Linear layers of BASELINE config 2 to the letter)
// The fp32 split-K instantiation of cf_igemm.hip (64 x 64 tiles, wave tile 32 x 32, both operands through 3-deep LDS rings, one barrier per
// 16-wide slab = per 8 MFMAs of a wave) runs 4096 x 512 x 512 in 46 us -- latency between barriers, not matrix work (13.7 us at the fp32 MFMA
// peak).  Here the structure of gemm_split_tile_kernel: 128 tokens x 64 columns per workgroup, a wave owns 64 tokens x 32 columns (two MFMA
// tiles), the token tile of a 32-wide stage staged once through LDS in operand order (conflict-free ds_read_b128, double-buffered, one
// barrier per stage = per 32 MFMAs of a wave), weight fragments straight from L2 in the fp32 packing of cf_pack_conv_weight
// ([K/16][N][16]: a lane's 16 bytes are W[n][16 slab + 8 kg + 4 (lane >> 5) .. + 3]), three stages ahead in registers.
// Per output element the arithmetic is that of the split-K instantiation with one workgroup per tile -- virtual chunks of 128 K values
// accumulated from zero on v_mfma_f32_32x32x2_f32 in slab order, k group 0 then 1, j = 0..3 (cf_mma_slab's operand convention), chunk sums
// added in chunk order, the same epilogue expression -- so the two agree BITWISE and the host may choose by the batch (this one: M % 128 == 0,
// one workgroup per tile) without touching batch invariance.
__global__ __launch_bounds__(256) void gemm_f32_tile_kernel(const GsArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][2][128][16];  // [buffer][16-wide slab of the stage][row][16 words]: 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ntn = g.N / 64;
  const int nt = blockIdx.x % ntn, mt = blockIdx.x / ntn;
  const int m0 = mt * 128, n0 = nt * 64 + wn * 32;
  const int nst = g.K >> 5;  // stages of 32 K values

  // staging: item j of this thread is float4 #q of row (tid >> 3) + 32 j of the tile's stage (k = 4 q .. 4 q + 3)
  const float* const asrc = gs_rows(g, n0) + (size_t)(m0 + (tid >> 3)) * g.K + (tid & 7) * 4;
  f32x4 rg[4][4];
  auto load_stage = [&](int st, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) rg[BUF][j] = *reinterpret_cast<const f32x4*>(asrc + (size_t)(32 * j) * g.K + st * 32);
  };
  auto store_stage = [&](int lbuf, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
    const int q = tid & 7, ks = q >> 2, ch = q & 3;   // 16-byte chunk ch of slab ks: k group ch >> 1, lane half ch & 1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (tid >> 3) + 32 * j;
      *reinterpret_cast<f32x4*>(&As[lbuf][ks][row][(ch ^ ((row >> 2) & 3)) << 2]) = rg[BUF][j];
    }
  };
  // weights: [K/16][N_pad][16] floats (g.acc_scale carries nothing here; N_pad == N is checked by the launch)
  const float* const wl = g.w + (size_t)(n0 + l31) * 16 + half * 4;
  const size_t kstride = (size_t)g.N * 16;  // floats between consecutive 16-wide slabs
  f32x4 rb[4][2][2];  // [ring slot][slab of the stage][k group]
  auto load_B = [&](int st, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* p = wl + (size_t)(st * 2 + s) * kstride;
      rb[BUF][s][0] = *reinterpret_cast<const f32x4*>(p);
      rb[BUF][s][1] = *reinterpret_cast<const f32x4*>(p + 8);
    }
  };
  f32x16 acc[2], tot[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = tot[mi][r] = 0.f;
  auto compute = [&](int lbuf, auto buf) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 a0[2], a1[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = wm * 64 + mi * 32 + l31;
        const int swz = (row >> 2) & 3;
        a0[mi] = *reinterpret_cast<const f32x4*>(&As[lbuf][s][row][(half ^ swz) << 2]);
        a1[mi] = *reinterpret_cast<const f32x4*>(&As[lbuf][s][row][((2 + half) ^ swz) << 2]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi][j], rb[BUF][s][0][j], acc[mi], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mi][j], rb[BUF][s][1][j], acc[mi], 0, 0, 0);
    }
  };
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>;
  using R3 = std::integral_constant<int, 3>;
  auto stage = [&](int st, auto cur, auto nxt, auto far, auto do_load, auto do_store) __attribute__((always_inline)) {
    if constexpr (decltype(do_load)::value) {
      load_stage(st + 3, far);
      load_B(st + 3, far);
    }
    __builtin_amdgcn_sched_barrier(0);  // (keep the prefetch ahead of the MFMAs)
    compute(st & 1, cur);
    if ((st & 3) == 3) {  // a virtual chunk of 128 K values is complete: fold it into the running sum, restart from zero
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        tot[mi] += acc[mi];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
      }
    }
    if constexpr (decltype(do_store)::value) store_stage((st + 1) & 1, nxt);
    __syncthreads();
  };
  constexpr std::true_type Y{};
  constexpr std::false_type NO{};
  load_stage(0, R0{});
  load_B(0, R0{});
  load_stage(1, R1{});
  load_B(1, R1{});
  load_stage(2, R2{});
  load_B(2, R2{});
  store_stage(0, R0{});
  __syncthreads();
  int st = 0;
  for (; st + 4 < nst; st += 4) {  // (K % 128 == 0: four stages per virtual chunk)
    stage(st, R0{}, R1{}, R3{}, Y, Y);
    stage(st + 1, R1{}, R2{}, R0{}, Y, Y);
    stage(st + 2, R2{}, R3{}, R1{}, Y, Y);
    stage(st + 3, R3{}, R0{}, R2{}, Y, Y);
  }
  stage(st, R0{}, R1{}, R3{}, Y, Y);   // the last chunk: one stage left to request, three to stage
  stage(st + 1, R1{}, R2{}, R0{}, NO, Y);
  stage(st + 2, R2{}, R3{}, R1{}, NO, Y);
  stage(st + 3, R3{}, R0{}, R2{}, NO, NO);

  // ---- epilogue: the expression of cf_igemm.hip's vector epilogue per element (+ bias, then GELU / + residual); lane holds column n of rows
  //      cf_acc_row(r, lane) of each of its tiles.  A wave's 32 lanes of a half store 128 contiguous bytes per row. ----
  const int n = n0 + l31;
  const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const size_t o = (size_t)(m0 + wm * 64 + mi * 32 + cf_acc_row(r, lane)) * g.N + n;
      float v = tot[mi][r] + bias;
      if (g.epilogue == CF_EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      if (g.epilogue == CF_EPI_RESIDUAL) v += g.res[o];
      g.out[o] = v;
    }
}

// W' = scale * W as hi = f16(W'), lo = f16(W' - hi) in MFMA-operand order [K/16][N/32][hi, lo][lane 64][4 words]:
// a lane's 16 bytes are the 8 halves of W'[n = tile*32 + (lane&31)][k = kstep*16 + (lane>>5)*8 + 0..7]
__global__ void pack_linear_f16x2_kernel(const float* __restrict__ w, int N, int K, float scale, unsigned* __restrict__ packed, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 3), ln = (int)((i >> 2) & 63), part = (int)((i >> 8) & 1);
  long r = i >> 9;
  const int ntiles = N / 32;
  const int n = (int)(r % ntiles) * 32 + (ln & 31);
  const int ks = (int)(r / ntiles);
  unsigned out = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = ks * 16 + (ln >> 5) * 8 + e * 2 + h;
    const float v = w[(long)n * K + k] * scale;  // exact: power of two
    const _Float16 hi = (_Float16)v;
    const _Float16 hv = part ? (_Float16)(v - (float)hi) : hi;
    out |= (unsigned)__builtin_bit_cast(unsigned short, hv) << (16 * h);
  }
  packed[i] = out;
}

}  // namespace

extern "C" int cf_pack_linear_weight_f16x2(const float* w, int n, int k, float scale, void* packed, cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_linear_weight_f16x2: null pointer");
  CF_REQUIRE(n > 0 && k > 0 && n % 64 == 0 && k % 128 == 0, "cf_pack_linear_weight_f16x2: N %d must be a multiple of 64, K %d of 128", n, k);
  int ex = 0;
  CF_REQUIRE(scale > 0.f && frexpf(scale, &ex) == 0.5f, "cf_pack_linear_weight_f16x2: scale %g is not a power of two", (double)scale);
  const long total = (long)n * k;  // 32-bit words: hi + lo half per weight
  hipLaunchKernelGGL(pack_linear_f16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, n, k, scale,
                     reinterpret_cast<unsigned*>(packed), total);
  CF_CHECK_LAUNCH("cf_pack_linear_weight_f16x2");
  return CF_OK;
}

// Called by cf_conv2d (cf_igemm.hip) for taps == 1 descriptors with bf16_mfma == CF_OPERAND_F16X2; the common argument checks have run.
int cf_gemm_split_geometry(const cf_conv_desc* d, int* tiles, long* bytes_per_part) {
  const long m = (long)d->batch * d->hout * d->wout;
  // (the weight packing fixes N % 64 == 0 for every form; the in-workgroup split -- 32 x 32 tiles -- takes M % 32 == 0, the others M % 64 == 0)
  const int mq = d->split_k == CF_SPLITK_IN_WORKGROUP ? 32 : 64;
  CF_REQUIRE(d->taps == 1 && m % mq == 0 && d->cout % 64 == 0 && d->cout_pad == d->cout && (d->c0 + d->c1) % 128 == 0,
             "cf_conv2d(1x1, f16x2): M %ld must be a multiple of %d, N %d of 64, K %d of 128", m, mq, d->cout, d->c0 + d->c1);
  *tiles = (int)(m / 64) * (d->cout / 64);
  const int V = (d->c0 + d->c1) / 128;
  *bytes_per_part = 64L * 64 * 4 * V / (d->split_k > 0 ? d->split_k : 1);  // V chunk sums of 16 KB per tile in all
  return CF_OK;
}

int cf_gemm_split_launch(const cf_conv_desc* d, hipStream_t stream) {
  int tiles = 0;
  long per = 0;
  const int rc = cf_gemm_split_geometry(d, &tiles, &per);
  if (rc != CF_OK) return rc;
  CF_REQUIRE(d->stride == 1 && !d->in_nchw && !d->out_nchw && d->c1 == 0 && d->prologue == CF_PRO_NONE && !d->stats_out &&
                 (d->ld_in0 == 0 || d->ld_in0 == d->c0) && (d->ld_out == 0 || d->ld_out == d->cout),
             "cf_conv2d(1x1, f16x2): dense single-input token GEMMs without prologue / statistics only");
  CF_REQUIRE(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_GELU || d->epilogue == CF_EPI_RESIDUAL,
             "cf_conv2d(1x1, f16x2): epilogues are none / GELU / residual");
  CF_REQUIRE(d->acc_scale > 0.f, "cf_conv2d(1x1, f16x2): acc_scale must be the inverse of the pack-time weight scale (got %g)", (double)d->acc_scale);
  CF_REQUIRE(!d->in0_alt || (d->alt_cout0 > 0 && d->alt_cout0 < d->cout && d->alt_cout0 % 128 == 0),
             "cf_conv2d(1x1, f16x2): in0_alt needs 0 < alt_cout0 < cout, a multiple of 128 (got %d of %d)", d->alt_cout0, d->cout);
  const int V = d->c0 / 128;
  if (d->split_k == CF_SPLITK_IN_WORKGROUP) {   // the chunks of a tile shared by the waves of one workgroup (same bits)
    const long m = (long)d->batch * d->hout * d->wout;
    CF_REQUIRE(V >= 1 && V <= GS_CHUNK_MAXV && m % 32 == 0,
               "cf_conv2d(1x1, f16x2, in-workgroup split): K %d must be a multiple of 128 up to %d, M a multiple of 32 (N %% 64 == 0: the packing)", d->c0, GS_CHUNK_MAXV * 128);
    GsArgs g;
    g.a_alt = d->in0_alt;
    g.n_alt = d->alt_cout0;
    g.a = d->in0;
    g.w = d->weight;
    g.bias = d->bias;
    g.res = d->res;
    g.out = d->out;
    g.M = (int)m;
    g.N = d->cout;
    g.K = d->c0;
    g.epilogue = d->epilogue;
    g.acc_scale = d->acc_scale;
    g.ws = nullptr;
    g.counters = nullptr;
    g.nsplit = 1;
    // (a 64 x 64-tile instantiation <2, 2> of the same kernel -- operands of half a chunk in flight -- was measured from 2048 rows up: 17.3 vs 15.2 us
    //  at 2048 x 512 x 512, 23.4 vs the token-tile kernel's 23.5 us at 4096 rows: no gain, not instantiated; the host sends large launches to the tile kernel)
    CF_LDS_ATTR((gemm_split_chunk_kernel<1, 1>), GS_CHUNK_MAXV * 4096);
    hipLaunchKernelGGL((gemm_split_chunk_kernel<1, 1>), dim3((unsigned)((g.M / 32) * (g.N / 32))), dim3(256), (size_t)V * 4096, stream, g);
    CF_CHECK_LAUNCH("cf_conv2d(1x1, f16x2, in-workgroup split)");
    return CF_OK;
  }
  const int nsplit = d->split_k >= 1 ? d->split_k : 1;
  CF_REQUIRE(V % nsplit == 0, "cf_conv2d(1x1, f16x2): split_k %d must divide K/128 = %d", nsplit, V);
  CF_REQUIRE(nsplit == 1 || (d->workspace && d->counters), "cf_conv2d(1x1, f16x2): split_k > 1 needs workspace and counters");
  GsArgs g;
  g.a_alt = d->in0_alt;
  g.n_alt = d->alt_cout0;
  g.a = d->in0;
  g.w = d->weight;
  g.bias = d->bias;
  g.res = d->res;
  g.out = d->out;
  g.M = (int)((long)d->batch * d->hout * d->wout);
  g.N = d->cout;
  g.K = d->c0;
  g.epilogue = d->epilogue;
  g.acc_scale = d->acc_scale;
  g.ws = d->workspace;
  g.counters = d->counters;
  g.nsplit = nsplit;
  // token tiles of 128 rows staged through LDS where the shape allows and K is not split (bitwise the same result: see the kernel);
  // the narrow form keeps a 128 x 64 grid at 256+ workgroups for N = 512
  // CF_GEMM_TILE (A/B only): 0 the untiled kernel, 1 (default) 128 x 64 tiles, 2 also 128 x 128 tiles where they still give 256 workgroups
  // (measured inside the step: 26.9 vs 29.6 us on 4096 x 512 x 1024 -- the narrow tile's 512 workgroups keep more loads in flight)
  static const int tile_mode = getenv("CF_GEMM_TILE") ? atoi(getenv("CF_GEMM_TILE")) : 1;
  if (nsplit == 1 && g.M % 128 == 0 && g.K % 64 == 0 && tile_mode != 0) {
    if (tile_mode == 2 && g.N % 128 == 0 && (long)(g.M / 128) * (g.N / 128) >= 256)
      hipLaunchKernelGGL(gemm_split_tile_kernel<2>, dim3((unsigned)((g.M / 128) * (g.N / 128))), dim3(256), 0, stream, g);
    else
      hipLaunchKernelGGL(gemm_split_tile_kernel<1>, dim3((unsigned)((g.M / 128) * (g.N / 64))), dim3(256), 0, stream, g);
    CF_CHECK_LAUNCH("cf_conv2d(1x1, f16x2, token tiles)");
    return CF_OK;
  }
  hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)(tiles * nsplit)), dim3(256), 0, stream, g);
  CF_CHECK_LAUNCH("cf_conv2d(1x1, f16x2)");
  return CF_OK;
}

// Called by cf_conv2d (cf_igemm.hip) for fp32 1x1 / Linear descriptors with split_k == 1 (one workgroup per output tile: large token
// matrices).  Returns CF_OK after launching, or 1 when the shape is not this kernel's (the caller then runs its split-K instantiation:
// the same bits).  CF_GEMM_F32_TILE=0 in the environment keeps every launch there (A/B).
int cf_gemm_f32_tile_try(const cf_conv_desc* d, hipStream_t stream) {
  static const bool on = !(getenv("CF_GEMM_F32_TILE") && atoi(getenv("CF_GEMM_F32_TILE")) == 0);
  const long m = (long)d->batch * d->hout * d->wout;
  const int k = d->c0;
  if (!on || d->taps != 1 || d->bf16_mfma != CF_OPERAND_F32 || d->split_k != 1 || d->c1 != 0 || d->stride != 1 || d->in_nchw || d->out_nchw ||
      d->prologue != CF_PRO_NONE || d->stats_out || d->stats_cpg || m % 128 || k % 128 || d->cout % 64 || d->cout_pad != d->cout ||
      (d->ld_in0 != 0 && d->ld_in0 != d->c0) || (d->ld_out != 0 && d->ld_out != d->cout) || d->pad_mode != CF_PAD_ZERO ||
      !(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_GELU || d->epilogue == CF_EPI_RESIDUAL))
    return 1;
  // 512 -> 512 (v / out projections) stays on the 64 x 64 instantiation: 29.4 us against 33.6 here at 4096 tokens (its 512 workgroups
  // against 256 one-wave-per-SIMD ones); every other Linear shape of the Transformer gains (profiles/r06_gemm_f32_tile_probe.txt:
  // 256 -> 512 19.0 -> 16.4 us, 512 -> 1024 52.7 -> 45.8, 1024 -> 512 54.7 -> 51.0).  A per-shape choice between bitwise-equal kernels.
  if (k == 512 && d->cout == 512) return 1;
  GsArgs g;
  g.a_alt = nullptr;   // (fp32 operands: one token matrix per launch)
  g.n_alt = 0;
  g.a = d->in0;
  g.w = d->weight;
  g.bias = d->bias;
  g.res = d->res;
  g.out = d->out;
  g.M = (int)m;
  g.N = d->cout;
  g.K = k;
  g.epilogue = d->epilogue;
  g.acc_scale = 1.f;
  g.ws = nullptr;
  g.counters = nullptr;
  g.nsplit = 1;
  hipLaunchKernelGGL(gemm_f32_tile_kernel, dim3((unsigned)((g.M / 128) * (g.N / 64))), dim3(256), 0, stream, g);
  CF_CHECK_LAUNCH("cf_conv2d(1x1, fp32 token tiles)");
  return CF_OK;
}
