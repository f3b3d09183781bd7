// Winograd F(2x2,3x3) convolution with split-half operands, 128 output channels per workgroup, software-pipelined (gfx950).
//
// The arithmetic is that of winograd_kernel<., H2 = true> (cf_winograd.hip): the 16 transform-domain GEMMs
//   M[xi,nu][tile][n] = sum_c V[xi,nu][tile][c] * U[xi,nu][c][n]
// run on v_mfma_f32_32x32x16_f16 with U and V as hi + lo IEEE halves (hi*hi + lo*hi + hi*lo, fp32 accumulation; U pre-split and
// scaled by a power of two at pack time, V split ONCE by the input transform, which writes it to LDS in operand format), i.e. 4/9 of the MFMA work of the direct
// split-half kernel (cf_split.hip) at its accuracy -- measured against fp64 it is the most accurate of the three 3x3 kernels.
//
// With the matrix work cut to 3 x 8-pass MFMAs per 16-channel slab and MFMA tile, the four-wave form is bound by everything
// else -- gather + GroupNorm/swish prologue, the input transform, the hi/lo conversion, barriers -- all of which it repeats for
// every 64-channel tile of the output.  This kernel is organised around that:
//   * a workgroup is EIGHT waves and owns an 8x16 output patch x 128 channels: wave = (xi, channel half); the halo patch is
//     gathered and transformed once for both halves, each thread handling half as many items;
//   * the patch buffer and V are double-buffered in LDS (121 KB, one workgroup per CU) and the slab loop is a four-stage pipeline
//     with ONE barrier per slab:  iteration k = { MMA(k) | prologue+store(k+2) | transform(k+1) | global loads(k+3) };
//   * the two waves that share a SIMD (w and w+4: the two channel halves) run the stages of an iteration in opposite order --
//     one starts with its MFMAs while the other transforms and gathers -- so the matrix pipe and the vector ALU of a SIMD are
//     busy at the same time (f16 MFMAs and VALU instructions of different waves do co-execute; the fp32 MFMAs of
//     cf_winograd.hip did not);
//   * weight fragments go global/L2 -> registers in MFMA-operand order; a fragment register is refilled for the next slab right
//     after its last use (a whole iteration of cover).
// Epilogue (nu axis in registers, xi axis through LDS, bias / residual / SFT, GroupNorm statistics of what was written) as in
// cf_winograd.hip, one 64-channel half per four-wave group.
//
// OP = CF_OPERAND_F16 / CF_OPERAND_BF16 (precision 'fp16' / 'bf16' of the network: BASELINE configs 3 and 5): the same kernel with
// SINGLE 16-bit operands -- U rounded once at pack time (the hi slot of the same fragment layout), V rounded by the transform, one
// MFMA per transform-domain product instead of three, half the weight-fragment registers and L2 traffic, no lo conversion.  fp32
// tensors, transform arithmetic, accumulation and epilogue are unchanged.
//
// BIO (round 6, cf_conv_desc.io_bf16; instantiated for OP = CF_OPERAND_BF16 only): the activations -- both concatenated inputs, the
// residual / SFT operands and the output -- are bf16 tensors in HBM.  The gather requests 8 bytes per item instead of 16 and widens in
// the patch store (a shift), the epilogue rounds once (RNE) after the GroupNorm partials were taken from the fp32 values.
#include <type_traits>

#include "cf_common.h"

// WS_ABLATE: timing-only ablation builds (tools/split_ab.sh); 0 / undefined in every product build.
// 1 no MFMAs, 2 no MMA stage at all, 4 no transform, 8 no prologue + patch store, 16 no epilogue, 32 no weight fetch

namespace {

constexpr int WS_TH = 8, WS_TW = 16;                 // output patch of a workgroup
constexpr int WS_PW = WS_TW + 2;                     // halo patch 10 x 18
constexpr int WS_NPIX = (WS_TH + 2) * WS_PW;         // 180
constexpr int WS_NT = 32;                            // Winograd tiles per patch = one MFMA row tile
constexpr int WS_THREADS = 512;
constexpr int WS_BN = 128;                           // output channels per workgroup
constexpr int WS_NI = 2;                             // 32-channel MFMA tiles per wave
constexpr int WS_PATCH_FLOATS = 256 * CF_LDK;        // 180 halo pixels padded to two gather rounds of 128 pixels (no guard on the store)
constexpr int WS_PS = WS_NT * CF_LDK + 4;            // 644 floats between positions of V (see cf_winograd.hip)
constexpr int WS_V_FLOATS = 16 * WS_PS;              // 10304
constexpr int WS_RLD = 36;                           // epilogue staging row: 32 channels + 4 pad
constexpr int WS_R_FLOATS = 8 * WS_NT * WS_RLD;      // 9216 per channel half
constexpr int WS_LDS_FLOATS = 2 * WS_PATCH_FLOATS + 2 * WS_V_FLOATS;
static_assert(2 * WS_R_FLOATS <= 2 * WS_V_FLOATS, "epilogue staging of both halves must fit the V buffers");

typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ws_f16x2 __attribute__((ext_vector_type(2)));

struct WsArgs {
  const float* in0;
  const float* in1;
  int c0, c1, cin, nchunks;
  int batch, h, w;
  int cout;
  int prologue, epilogue;
  const float* pro_scale;
  const float* pro_shift;
  const float* weight;  // [16 pos][nchunks][cout/32][hi, lo][64 lanes][4 words]  (cf_pack_conv_weight_winograd_f16x2)
  const float* bias;
  const float* res;
  const float* sft_scale;
  float sft_w;
  float acc_scale;
  const float* act_scale;  // [batch][2] (s, 1/s) powers of two for un-normalised inputs (PRO NONE / LEAKY), or null
  float* out;
  double* stats_out;
  int stats_cpg, nparts;
  int tiles_x, tiles_per_img, ntn;
};

// PRO = the prologue (enum cf_prologue) as a template parameter: with a switch inside the slab loop hipcc's wait-count pass merged the
// branches into s_waitcnt vmcnt(0) before the patch store, i.e. every iteration waited for the weight fragments it had just requested.
template <int PRO, int OP = CF_OPERAND_F16X2, bool BIO = false>
__global__ __launch_bounds__(WS_THREADS, 1) void wsplit_kernel(const WsArgs a) {
  constexpr int NI = WS_NI;
  constexpr int NPART = OP == CF_OPERAND_F16X2 ? 2 : 1;  // weight-fragment parts a lane keeps: hi + lo, or the single rounded operand
  constexpr int APT = 2;  // float4 gather items per thread: 256 pixel slots x 4 quads / 512 threads
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const patch0 = smem;
  float* const V0 = smem + 2 * WS_PATCH_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int xi = wave & 3;
  const int nh = wave >> 2;  // channel half of this wave (waves w and w + 4 share a SIMD)
  const int gtid = tid & 255;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  int bid = blockIdx.x;
  {  // XCD-contiguous tile order (see cf_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nt = bid % a.ntn;
  const int mt = bid / a.ntn;
  const int n0 = nt * WS_BN + nh * 64;  // first output channel of this wave's half
  const int b = mt / a.tiles_per_img;
  const int rt = mt - b * a.tiles_per_img;
  const int tyw = rt / a.tiles_x;
  const int y0 = tyw * WS_TH;
  const int x0 = (rt - tyw * a.tiles_x) * WS_TW;
  const int n = a.nchunks;

  // ---- gather: item j of this thread is float4 #k4 of halo pixel p = (tid >> 2) + 128 j ----
  const int k4 = tid & 3;
  int pix[APT];
#pragma unroll
  for (int j = 0; j < APT; ++j) {
    const int p = (tid >> 2) + 128 * j;
    int v = -1;
    if (p < WS_NPIX) {
      const int hy = p / WS_PW;
      const int hx = p - hy * WS_PW;
      const int iy = y0 - 1 + hy;
      const int ix = x0 - 1 + hx;
      if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) v = (b * a.h + iy) * a.w + ix;
    }
    pix[j] = v;
  }
  constexpr bool affine = PRO == CF_PRO_AFFINE || PRO == CF_PRO_AFFINE_SWISH;
  // range scale of an un-normalised input (cf_conv_desc.act_scale): powers of two, so x * s and acc / s are exact; 1 when unused
  float act_s = 1.f, act_is = 1.f;
  if (!affine && a.act_scale) {
    act_s = a.act_scale[2 * b];
    act_is = a.act_scale[2 * b + 1];
  }
  const float act_s02 = 0.2f * act_s;  // LeakyReLU slope folded with the scale: fl(y * (0.2 s)) == fl(0.2 y) * s
  const float* const tab_sc = affine ? a.pro_scale + (size_t)b * a.cin : a.weight;  // (any valid address when unused)
  const float* const tab_sh = affine ? a.pro_shift + (size_t)b * a.cin : a.weight;
  f32x4 rsc, rsh;
  std::conditional_t<BIO, cf_u32x2, f32x4> ra[APT];   // (bf16 storage: the raw 8 bytes ride through the pipeline, widened at the store)
  // unconditional loads from clamped addresses; out-of-image items are zeroed at the store (see cf_winograd.hip)
  auto load_A = [&](int chunk) __attribute__((always_inline)) {
    const int c = chunk * CF_BK + k4 * 4;
    rsc = *reinterpret_cast<const f32x4*>(tab_sc + (affine ? c : 0));
    rsh = *reinterpret_cast<const f32x4*>(tab_sh + (affine ? c : 0));
    const bool first = c < a.c0;
    const float* src = first ? a.in0 : a.in1;
    const int cs = first ? a.c0 : a.c1;
    const int cc = first ? c : c - a.c0;
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      const int pj = pix[j] < 0 ? 0 : pix[j];
      if constexpr (BIO) ra[j] = *reinterpret_cast<const cf_u32x2*>(reinterpret_cast<const unsigned short*>(src) + (size_t)pj * cs + cc);
      else ra[j] = *reinterpret_cast<const f32x4*>(src + (size_t)pj * cs + cc);
    }
  };
  auto store_patch = [&](float* patch) __attribute__((always_inline)) {
    const f32x4 sc = rsc, sh = rsh;
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      const int p = (tid >> 2) + 128 * j;  // (p >= 180: padding rows of the patch buffer, written as zeros -- no branch)
      const bool valid = pix[j] >= 0;
      f32x4 v;
      if constexpr (BIO) v = cf_bf16x4_widen(ra[j]);
      else v = ra[j];
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

  // ---- input transform: one item (tile, channel quad, xi row) per thread ----
  // B^T d along rows:  xi0 = r0 - r2, xi1 = r1 + r2, xi2 = r2 - r1, xi3 = r1 - r3, each as q + s*p (one FMA per value, exact);
  // (.) B along columns: nu0 = t0 - t2, nu1 = t1 + t2, nu2 = t2 - t1, nu3 = t1 - t3
  const int t_c4 = tid & 3, t_xi = (tid >> 2) & 3, t_tile = tid >> 4;
  const int t_q = t_xi == 0 ? 0 : (t_xi == 2 ? 2 : 1);
  const int t_p = t_xi == 2 ? 1 : (t_xi == 3 ? 3 : 2);
  const float t_s = t_xi == 1 ? 1.f : -1.f;
  const int t_off_q = ((2 * (t_tile >> 3) + t_q) * WS_PW + 2 * (t_tile & 7)) * CF_LDK + t_c4 * 4;
  const int t_off_p = ((2 * (t_tile >> 3) + t_p) * WS_PW + 2 * (t_tile & 7)) * CF_LDK + t_c4 * 4;
  const int t_off_v = t_xi * 4 * WS_PS + t_tile * CF_LDK + t_c4 * 2;  // (floats) this item's two words of the row's hi half; lo: + 8
  auto transform = [&](const float* patch, float* V) __attribute__((always_inline)) {
    f32x4 t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 dq = *reinterpret_cast<const f32x4*>(patch + t_off_q + c * CF_LDK);
      const f32x4 dp = *reinterpret_cast<const f32x4*>(patch + t_off_p + c * CF_LDK);
#pragma unroll
      for (int e = 0; e < 4; ++e) t[c][e] = __fmaf_rn(dp[e], t_s, dq[e]);
    }
    // (.) B along columns, then the operand conversion -- ONCE per element, here: V rows hold the slab's 16 channels as
    // [hi: 16 halves | lo: 16 halves] (the 64 bytes the fp32 values took), so a lane's A fragment is two 16-byte reads and the two
    // channel-half waves that share it convert nothing (48 of the 161 VALU instructions of their slab loop went there).  Single-operand
    // modes: the rounded value in the hi slot.
    const f32x4 v4[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
    typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
    float* vo = V + t_off_v;
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      if constexpr (OP == CF_OPERAND_F16X2) {
        float h0, l0, h1, l1;
        cf_split_pair(v4[nu][0], v4[nu][1], h0, l0);
        cf_split_pair(v4[nu][2], v4[nu][3], h1, l1);
        *reinterpret_cast<ws_f32x2*>(vo + nu * WS_PS) = ws_f32x2{h0, h1};
        *reinterpret_cast<ws_f32x2*>(vo + nu * WS_PS + 8) = ws_f32x2{l0, l1};
      } else if constexpr (OP == CF_OPERAND_F16) {
        const ws_f16x2 a0 = {(_Float16)v4[nu][0], (_Float16)v4[nu][1]}, a1 = {(_Float16)v4[nu][2], (_Float16)v4[nu][3]};
        *reinterpret_cast<ws_f32x2*>(vo + nu * WS_PS) = ws_f32x2{__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1)};
      } else {
        typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
        const ws_bf16x2 a0 = {(__bf16)v4[nu][0], (__bf16)v4[nu][1]}, a1 = {(__bf16)v4[nu][2], (__bf16)v4[nu][3]};
        *reinterpret_cast<ws_f32x2*>(vo + nu * WS_PS) = ws_f32x2{__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1)};
      }
    }
  };

  // ---- MFMA stage: wave (xi, nh) owns positions (xi, 0..3) x 64 channels ----
  f32x16 acc[4][NI];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][ni][r] = 0.f;
  const size_t pos_stride = (size_t)a.nchunks * a.cout * CF_BK;
  const float* const wlane = a.weight + (size_t)(xi * 4) * pos_stride + (size_t)(n0 / 32) * 512 + lane * 4;
  f32x4 bq[4][NI][NPART];  // [nu][n tile][hi, lo]
  auto load_B = [&](int chunk, int nu) __attribute__((always_inline)) {
    const float* wc = wlane + (size_t)chunk * a.cout * CF_BK + nu * pos_stride;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int part = 0; part < NPART; ++part) bq[nu][ni][part] = *reinterpret_cast<const f32x4*>(wc + ni * 512 + part * 256);
  };
  const int a_off = (xi * 4) * WS_PS + l31 * CF_LDK + half * 4;
  // A fragments: this lane's 8 channels of the slab (row = tile l31, channels half*8 .. +7) of position (xi, nu), already in operand
  // format: [0] = the 8 hi halves (or the single rounded operand), [1] = the 8 lo halves; the reads for position nu + 1 are issued
  // before the MFMAs of nu (the LDS latency of a position was exposed four times per slab)
  f32x4 va[2][NPART];
  auto read_A = [&](const float* V, int nu) __attribute__((always_inline)) {
    va[nu & 1][0] = *reinterpret_cast<const f32x4*>(V + a_off + nu * WS_PS);
    if constexpr (NPART == 2) va[nu & 1][1] = *reinterpret_cast<const f32x4*>(V + a_off + nu * WS_PS + 8);
  };
  auto mma = [&](int nu) __attribute__((always_inline)) {
    if constexpr (OP != CF_OPERAND_F16X2) {  // single 16-bit operands: one MFMA per product
      const f32x4 as = va[nu & 1][0];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if constexpr (OP == CF_OPERAND_F16) {
          acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, as), __builtin_bit_cast(ws_f16x8, bq[nu][ni][0]),
                                                               acc[nu][ni], 0, 0, 0);
        } else {
          typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
          acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ws_bf16x8, as), __builtin_bit_cast(ws_bf16x8, bq[nu][ni][0]),
                                                                acc[nu][ni], 0, 0, 0);
        }
      }
      return;
    } else {
      const f32x4 ah = va[nu & 1][0], al = va[nu & 1][NPART - 1];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, al), __builtin_bit_cast(ws_f16x8, bq[nu][ni][0]),
                                                             acc[nu][ni], 0, 0, 0);
        acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, ah), __builtin_bit_cast(ws_f16x8, bq[nu][ni][NPART - 1]),
                                                             acc[nu][ni], 0, 0, 0);
        acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, ah), __builtin_bit_cast(ws_f16x8, bq[nu][ni][0]),
                                                             acc[nu][ni], 0, 0, 0);
      }
    }
  };
  auto mma_stage = [&](const float* V, int next_chunk) __attribute__((always_inline)) {
    read_A(V, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      if (nu < 3) read_A(V, nu + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(nu);
      load_B(next_chunk, nu);  // refill for the next slab (clamped on the last one): a whole iteration of cover
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto feed_stage = [&](int k) __attribute__((always_inline)) {  // store(k+2), transform(k+1), loads(k+3)
    const int buf = k & 1;
    // The store comes first: it retires the sixteen registers of the activation prefetch (values + GroupNorm rows) before the transform
    // takes its thirty-two (the two stages touch different buffers: patch[buf] / patch[buf ^ 1] -> V[buf ^ 1]).  In the other order the
    // GN-swish instantiations spilled ten registers: 0.877 -> 0.841 ms on 128->128 @256^2, 0.598 -> 0.543 ms on 64->128 (same bits).
    if (k + 2 < n) store_patch(patch0 + buf * WS_PATCH_FLOATS);
    __builtin_amdgcn_sched_barrier(0);
    if (k + 1 < n) transform(patch0 + (buf ^ 1) * WS_PATCH_FLOATS, V0 + (buf ^ 1) * WS_V_FLOATS);
    load_A(k + 3 < n ? k + 3 : n - 1);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- pipeline fill + slab loop, one straight-line copy per role.  One barrier per slab; the two waves of a SIMD run the stages of
  // an iteration in opposite order.  The copies differ in WHERE the first weight fragments are requested: the loads still in flight at
  // the loop entry must be in the order the loop body leaves them in (MMA-first: weights then activations; feed-first: activations
  // then weights) -- hipcc's wait counts at the top of the loop are the merge of both states, and with the wrong order in front of
  // the loop every iteration waited for the weight fragments it had just requested (vmcnt(3) where vmcnt(16) is meant) ----
  auto all_B = [&]() __attribute__((always_inline)) {
    load_B(0, 0);
    load_B(0, 1);
    load_B(0, 2);
    load_B(0, 3);
  };
  auto fill = [&](auto mma_first) __attribute__((always_inline)) {
    constexpr bool MF = decltype(mma_first)::value;
    load_A(0);
    if (MF) all_B();
    __builtin_amdgcn_sched_barrier(0);
    store_patch(patch0);
    load_A(n > 1 ? 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    transform(patch0, V0);
    if (n > 1) store_patch(patch0 + WS_PATCH_FLOATS);
    load_A(n > 2 ? 2 : n - 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!MF) all_B();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };
  if (nh == 0) {
    fill(std::true_type{});
    for (int k = 0; k < n; ++k) {
      mma_stage(V0 + (k & 1) * WS_V_FLOATS, k + 1 < n ? k + 1 : k);
      feed_stage(k);
      __syncthreads();
    }
  } else {
    fill(std::false_type{});
    for (int k = 0; k < n; ++k) {
      feed_stage(k);
      mma_stage(V0 + (k & 1) * WS_V_FLOATS, k + 1 < n ? k + 1 : k);
      __syncthreads();
    }
  }

  // ---- to the output domain (both channel halves side by side, each four-wave group in its own staging region) ----
  float* const R = V0 + nh * WS_R_FLOATS;  // [(xi*2 + bb)][tile][WS_RLD]
  const int e_n4 = gtid & 7;
  auto to_output = [&](int pass, f32x4(&o)[4]) __attribute__((always_inline)) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][pass][r], m1 = acc[1][pass][r], m2 = acc[2][pass][r], m3 = acc[3][pass][r];
      const int row = cf_acc_row(r, lane);
      R[((xi * 2 + 0) * WS_NT + row) * WS_RLD + l31] = (m0 + m1) + m2;  // nu axis: R[xi][0] = M0 + M1 + M2
      R[((xi * 2 + 1) * WS_NT + row) * WS_RLD + l31] = (m1 - m2) - m3;  //          R[xi][1] = M1 - M2 - M3
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int it = gtid + k * 256;
      const int e_bb = (it >> 3) & 1, e_tile = it >> 4;
      f32x4 x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(R + ((q * 2 + e_bb) * WS_NT + e_tile) * WS_RLD + e_n4 * 4);
      o[k * 2 + 0] = (x[0] + x[1]) + x[2];  // xi axis: Y[0][bb] = R0 + R1 + R2 ; Y[1][bb] = R1 - R2 - R3
      o[k * 2 + 1] = (x[1] - x[2]) - x[3];
    }
  };

  // ---- epilogue: residual / SFT operands of both passes requested first, then bias / residual / SFT, 16-byte stores, statistics ----
  unsigned offs[NI][4];
  f32x4 r0[NI][4], r1[NI][4];
  // offsets: one base per thread, the (pass, k, aa) variants differ by wave-uniform amounts (k: 16 tiles = 4 pixel rows, aa: one row)
  const unsigned e_rowc = (unsigned)a.w * (unsigned)a.cout;
  const unsigned e_base = (((unsigned)b * a.h + (y0 + 2 * ((gtid >> 4) >> 3))) * a.w + (x0 + 2 * ((gtid >> 4) & 7) + ((gtid >> 3) & 1))) * (unsigned)a.cout +
                          (n0 + e_n4 * 4);
#pragma unroll
  for (int pass = 0; pass < NI; ++pass)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        const unsigned off = e_base + (unsigned)(k * 4 + aa) * e_rowc + pass * 32;
        offs[pass][k * 2 + aa] = off;
        r0[pass][k * 2 + aa] = r1[pass][k * 2 + aa] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BIO) {
          if (a.epilogue == CF_EPI_RESIDUAL || a.epilogue == CF_EPI_SFT) r0[pass][k * 2 + aa] = cf_load4_bf16(a.res, off);
          if (a.epilogue == CF_EPI_SFT) r1[pass][k * 2 + aa] = cf_load4_bf16(a.sft_scale, off);
        } else {
          if (a.epilogue == CF_EPI_RESIDUAL || a.epilogue == CF_EPI_SFT) r0[pass][k * 2 + aa] = *reinterpret_cast<const f32x4*>(a.res + off);
          if (a.epilogue == CF_EPI_SFT) r1[pass][k * 2 + aa] = *reinterpret_cast<const f32x4*>(a.sft_scale + off);
        }
      }
    }
#pragma unroll
  for (int pass = 0; pass < NI; ++pass) {
    f32x4 o[4];
    to_output(pass, o);
    const int nn = n0 + pass * 32 + e_n4 * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + nn);
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const float acc_s = a.acc_scale * act_is;  // (a product of powers of two: exact)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = o[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * acc_s + bias4[e];  // (a power of two: exact)
      if (a.epilogue == CF_EPI_RESIDUAL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r0[pass][i][e];
      } else if (a.epilogue == CF_EPI_SFT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r0[pass][i][e] + a.sft_w * (r0[pass][i][e] * r1[pass][i][e] + v[e]);
      }
      if constexpr (BIO) cf_store4_bf16(a.out, offs[pass][i], v);   // (rounded here, once; the statistics below see the fp32 values)
      else *reinterpret_cast<f32x4*>(a.out + offs[pass][i]) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ssum[e] += v[e];
        ssq[e] += v[e] * v[e];
      }
    }
    if (a.stats_out) {
      // GroupNorm statistics of the values just written (fp64 partials, fixed shuffle order): one partial per
      // (image, group, output patch, xi wave) -- nparts = tiles_per_img * 4, as the four-wave kernel writes them
      const int cpg = a.stats_cpg;  // >= 4 here (cout >= 128)
      double d0 = ((double)ssum[0] + ssum[1]) + ((double)ssum[2] + ssum[3]);
      double q0 = ((double)ssq[0] + ssq[1]) + ((double)ssq[2] + ssq[3]);
      for (int o2 = 8; o2 < 64; o2 <<= 1) {  // the (tile, bb) items of this wave: lanes with the same channel quad
        d0 += __shfl_xor(d0, o2, 64);
        q0 += __shfl_xor(q0, o2, 64);
      }
      for (int o2 = 1; o2 * 4 < cpg; o2 <<= 1) {  // adjacent channel quads of one group (cpg >= 8)
        d0 += __shfl_xor(d0, o2, 64);
        q0 += __shfl_xor(q0, o2, 64);
      }
      if ((lane >> 3) == 0 && (nn % cpg) == 0) {
        const size_t pidx = (size_t)rt * 4 + xi;
        const int ng = a.cout / cpg;
        double* op = a.stats_out + (((size_t)b * ng + nn / cpg) * a.nparts + pidx) * 2;
        op[0] = d0;
        op[1] = q0;
      }
    }
  }
}

}  // namespace

// Called by cf_winograd_launch (cf_winograd.hip) for split-half Winograd descriptors this kernel covers; argument checks have run there.
bool cf_wsplit_covers(const cf_conv_desc* d) {
  return d->winograd && (d->bf16_mfma == CF_OPERAND_F16X2 || d->bf16_mfma == CF_OPERAND_F16 || d->bf16_mfma == CF_OPERAND_BF16) &&
         d->split_k < 1 && d->cout % WS_BN == 0 && d->cout_pad == d->cout &&
         d->hout % WS_TH == 0 && d->wout % WS_TW == 0 && (long)d->hout * d->wout >= 32 * 32 && (d->stats_cpg == 0 || d->stats_cpg >= 4);
}

int cf_wsplit_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query) {
  WsArgs a;
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
  a.stats_cpg = d->stats_cpg > 0 ? d->stats_cpg : 4;
  a.tiles_x = d->wout / WS_TW;
  a.tiles_per_img = a.tiles_x * (d->hout / WS_TH);
  a.nparts = a.tiles_per_img * 4;
  a.ntn = d->cout / WS_BN;
  if (parts_query) {
    *parts_query = a.nparts;
    return CF_OK;
  }
  constexpr size_t lds = WS_LDS_FLOATS * sizeof(float);
  // (cf_device_init sets the dynamic-LDS attribute of the twelve instantiations on each device)
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_NONE>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE_SWISH>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_LEAKY>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_NONE, CF_OPERAND_F16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE, CF_OPERAND_F16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE_SWISH, CF_OPERAND_F16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_LEAKY, CF_OPERAND_F16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_NONE, CF_OPERAND_BF16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE, CF_OPERAND_BF16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE_SWISH, CF_OPERAND_BF16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_LEAKY, CF_OPERAND_BF16>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_NONE, CF_OPERAND_BF16, true>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE, CF_OPERAND_BF16, true>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_AFFINE_SWISH, CF_OPERAND_BF16, true>), lds);
  CF_LDS_ATTR((wsplit_kernel<CF_PRO_LEAKY, CF_OPERAND_BF16, true>), lds);
  const dim3 grid(a.tiles_per_img * d->batch * a.ntn), block(WS_THREADS);
  auto launch_op = [&](auto op, auto bio) {
    constexpr int OP = decltype(op)::value;
    constexpr bool BIO = decltype(bio)::value;
    switch (d->prologue) {
      case CF_PRO_AFFINE: hipLaunchKernelGGL((wsplit_kernel<CF_PRO_AFFINE, OP, BIO>), grid, block, lds, stream, a); break;
      case CF_PRO_AFFINE_SWISH: hipLaunchKernelGGL((wsplit_kernel<CF_PRO_AFFINE_SWISH, OP, BIO>), grid, block, lds, stream, a); break;
      case CF_PRO_LEAKY: hipLaunchKernelGGL((wsplit_kernel<CF_PRO_LEAKY, OP, BIO>), grid, block, lds, stream, a); break;
      default: hipLaunchKernelGGL((wsplit_kernel<CF_PRO_NONE, OP, BIO>), grid, block, lds, stream, a); break;
    }
  };
  CF_REQUIRE(!d->io_bf16 || d->bf16_mfma == CF_OPERAND_BF16, "cf_conv2d(winograd, 8 waves): bf16 tensors (io_bf16) go with CF_OPERAND_BF16 operands");
  if (d->bf16_mfma == CF_OPERAND_F16) launch_op(std::integral_constant<int, CF_OPERAND_F16>{}, std::false_type{});
  else if (d->bf16_mfma == CF_OPERAND_BF16 && d->io_bf16) launch_op(std::integral_constant<int, CF_OPERAND_BF16>{}, std::true_type{});
  else if (d->bf16_mfma == CF_OPERAND_BF16) launch_op(std::integral_constant<int, CF_OPERAND_BF16>{}, std::false_type{});
  else launch_op(std::integral_constant<int, CF_OPERAND_F16X2>{}, std::false_type{});
  CF_CHECK_LAUNCH("cf_conv2d(winograd f16x2)");
  return CF_OK;
}
