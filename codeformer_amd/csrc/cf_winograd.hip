// Winograd F(2x2, 3x3) convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// For a 3x3 stride-1 convolution (vqgan_arch.py:147-164 ResBlock convs, codeformer_arch.py:142-156 fusion convs) each 2x2
// block of outputs is   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   with d the 4x4 input patch, g the 3x3 kernel and
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// so the contraction over input channels becomes 16 independent GEMMs (one per position (xi, nu) of the 4x4 transform domain)
//   M[xi,nu][tile][n] = sum_c V[xi,nu][tile][c] * U[xi,nu][c][n]
// with 16 instead of 36 multiplies per 2x2 outputs: 2.25x fewer MFMA cycles than the direct implicit GEMM (cf_igemm.hip),
// whose main loop already runs at the clock-limited MFMA rate.  All arithmetic is fp32; this is an exact algebraic identity
// evaluated in a different order.  Measured against fp64 its error is BELOW the direct kernel's (1-6e-6 vs 2-14e-6 on O(5)
// outputs: fewer products are summed per output), so it is used for every eligible 3x3 stride-1 convolution of the network,
// encoder included (logits 4.7e-6 / indices exact against the reference).
//
// Work decomposition (256 threads = 4 waves, two workgroups per CU):
//   * a workgroup owns an 8x16 output patch of ONE image (32 Winograd tiles) x 64 output channels;
//   * per 16-channel K slab the 10x18 halo patch is gathered into LDS once -- GroupNorm-apply/swish or LeakyReLU prologue,
//     channel concat and zero padding resolved in the gather exactly as in cf_igemm.hip -- and transformed IN LDS to
//     V[16 positions][32 tiles][16 ch] (item = (tile, channel quad, xi row): 8 ds_read_b128, 8 vector ops, 4 ds_write_b128);
//   * wave xi (0..3) owns the four positions (xi, 0..3): per position a 32(tiles) x 64(n) accumulator = 2 MFMA tiles, 128
//     accumulator registers per lane in total.  A fragments come from V in LDS; B fragments (the transformed weights, packed in
//     MFMA-operand order so that a fragment load is one contiguous 1 KB block) are used by exactly one wave and go
//     global/L2 -> registers directly;
//   * two barriers per slab (patch visible / V visible);
//   * epilogue: each wave contracts its own nu axis in registers (R[xi][b] = M[xi][.] A), the xi axis is contracted across waves
//     through LDS, then item = (tile, output column, channel quad) holds two output pixels as float4s and applies bias /
//     residual / SFT, 16-byte stores, and the GroupNorm statistics of what it wrote (fp64 partials, fixed shuffle order).
//
// What the measurements said (interleaved ablations, tools/wino_ab.py; 128->128 @ 256x256 x16: direct 2.31 ms, this 1.58 ms):
//   * fp32 MFMA and fp32 VALU do NOT overlap on this part -- the fp32 matrix peak equals the fp32 vector peak because it is the
//     same FMA hardware.  MFMA-only, everything-else-only and the full kernel came out additive (1.35 + 0.55 = 1.90) even in a
//     ping-pong variant that pins one MFMA wave and one VALU wave on every SIMD; so every VALU instruction of the
//     gather / transform / epilogue is paid for in full, and what helped was removing instructions and stalls:
//   * the per-lane add / sub choice of the transform compiled into divergent branches with an LDS wait in each (now sign
//     multipliers, branch-free); the GroupNorm scale / shift rows loaded inside the gather-store put an s_waitcnt vmcnt(0) there
//     that also drained the weight fragments just requested (now fetched a slab ahead with the activations); the guard on the
//     third gather item was a branch (the patch buffer is padded to 192 pixels instead); touching the prefetched activations
//     (zero-fill select) right after the load made the wave wait for HBM (the select moved to the store): 2.45 -> 1.58 ms;
//   * producer / consumer wave specialisation (4 MFMA waves + 4 gather / transform waves) was slower (2.7 ms): four waves alone
//     cannot hide the LDS / transcendental latencies of the A-side work.
#include <type_traits>

#include "cf_common.h"

namespace {

constexpr int WG_TH = 8, WG_TW = 16;                      // output patch of a workgroup
constexpr int WG_PH = WG_TH + 2, WG_PW = WG_TW + 2;       // halo patch 10 x 18
constexpr int WG_NPIX = WG_PH * WG_PW;                    // 180
constexpr int WG_NT = (WG_TH / 2) * (WG_TW / 2);          // 32 Winograd tiles = one MFMA row tile
constexpr int WG_NI = 2;                                  // 32-channel MFMA tiles per consumer wave
constexpr int WG_BN = WG_NI * 32;                         // output channels per workgroup (64)
constexpr int WG_PATCH_FLOATS = 256 * CF_LDK;             // 5120: 180 halo pixels, padded to a whole number of gather rounds
                                                          // (3 x 64 or 2 x 128 pixels) so the gather-store needs no guard
// Floats between consecutive positions of V: 32 rows of 20 floats + 4 floats of padding.  4*PS*4 B = 64 (mod 256), so the four
// xi rows written by neighbouring lanes of the transform land in distinct 64-byte bank groups (unpadded they all aliased).
constexpr int WG_PS = WG_NT * CF_LDK + 4;                 // 644
constexpr int WG_V_FLOATS = 16 * WG_PS;                   // 10304
constexpr int WG_RLD = 36;                                // epilogue staging row: 32 channels + 4 pad (144 B, 16 B aligned)
constexpr int WG_R_FLOATS = 8 * WG_NT * WG_RLD;           // 9216 <= WG_V_FLOATS (staging reuses the V region)
static_assert(WG_R_FLOATS <= WG_V_FLOATS, "epilogue staging must fit the V region");

struct WinoArgs {
  const float* in0;
  const float* in1;
  int c0, c1, cin, nchunks;
  int batch, h, w;
  int cout, cout_pad;
  int prologue, epilogue;
  const float* pro_scale;
  const float* pro_shift;
  const float* weight;  // [16 pos][nchunks][cout_pad][16]
  const float* bias;
  const float* res;
  const float* sft_scale;
  float sft_w;
  float acc_scale;  // H2: inverse of the pack-time weight scale (a power of two); 1 otherwise
  const float* act_scale;  // H2: [batch][2] (s, 1/s) powers of two for un-normalised inputs (prologue NONE / LEAKY), or null
  float* out;
  double* stats_out;
  int stats_cpg, nparts;
  int tiles_x, tiles_per_img, ntn;
};
// split-K instantiation (layers of at most 32x32 pixels per face, see cf_common.h): workspace, ticket counters, split count
struct WinoArgsSK : WinoArgs {
  float* ws;
  unsigned* counters;
  int nsplit;
};

__device__ __forceinline__ f32x4 v4add(f32x4 a, f32x4 b) { return a + b; }
__device__ __forceinline__ f32x4 v4sub(f32x4 a, f32x4 b) { return a - b; }

// One workgroup (four waves) computes one 8x16 output patch x 64 channels; see the file header.  Two measured variants are not
// shipped (git history): a "ping-pong" pair of groups per workgroup running two barrier slots apart so that every SIMD always has
// one MFMA wave and one VALU / LDS wave (1.90 ms against 1.58 ms on 128->128 @256x256x16 -- the two streams do not co-execute,
// SQ_VALU_MFMA_COEXEC_CYCLES = 0), and an eight-wave workgroup owning 128 channels (1.65 ms).
// SK = true: the K range is cut into virtual chunks of CF_SK_SLABS slabs.  At the end of every chunk the accumulators are taken to the
// OUTPUT domain (the first half of the epilogue: nu axis in registers, xi axis through LDS -- 32 values per thread instead of 128) and
// the chunk sums are added in chunk order, either in registers (one workgroup per patch) or -- `nsplit` workgroups sharing a patch --
// through the workspace by the workgroup that draws the last ticket (cf_splitk_park / cf_splitk_finish): the same bits for every nsplit.
// The running sum lives in LDS (32 KB behind the V buffer), not in registers: next to the 128 Winograd-domain accumulators a second
// register accumulator spilled into the main loop (+36 %).
// H2 = true ("split halves", the scheme of cf_split.hip in the Winograd domain): the 16 GEMMs run on v_mfma_f32_32x32x16_f16 with both
// operands as hi + lo IEEE halves -- U pre-split at pack time (scaled by a power of two so that the lo halves stay normal), V split
// once by the input transform, which writes it to LDS in operand format -- and
// hi*hi + lo*hi + hi*lo accumulated in fp32: 3 MFMAs of 8 passes per 16-channel slab and MFMA tile instead of 8 of 16 passes.
// Gather, transform, epilogue and the split-K protocol are shared with the fp32 form.
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wg_f16x2 __attribute__((ext_vector_type(2)));

template <bool SK, bool H2>
__global__ __launch_bounds__(256, 2) void winograd_kernel(const std::conditional_t<SK, WinoArgsSK, WinoArgs> a) {
  constexpr int NI = WG_NI;
  constexpr int GT = 256;                                        // threads
  constexpr int APT = (WG_NPIX * 4 + GT - 1) / GT;               // float4 gather items per thread (3)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int xi = wave & 3;
  const int gtid = tid;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  float* const patch = smem;
  float* const V = patch + WG_PATCH_FLOATS;

  int bid = blockIdx.x;
  {  // XCD-contiguous tile order (see cf_igemm.hip)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  [[maybe_unused]] int split = 0, sk_tile = 0;
  if constexpr (SK) {
    split = bid % a.nsplit;
    bid /= a.nsplit;
    sk_tile = bid;
  }
  const int nt = bid % a.ntn;
  const int mt = bid / a.ntn;  // this workgroup's output patch
  const int n0 = nt * WG_BN;
  const int b = mt / a.tiles_per_img;
  const int rt = mt - b * a.tiles_per_img;
  const int tyw = rt / a.tiles_x;
  const int y0 = tyw * WG_TH;
  const int x0 = (rt - tyw * a.tiles_x) * WG_TW;
  int n = a.nchunks, kb = 0;  // this workgroup's K slabs: [kb, kb + n)
  if constexpr (SK) {
    n = (a.nchunks / CF_SK_SLABS / a.nsplit) * CF_SK_SLABS;  // whole virtual chunks (host-checked)
    kb = split * n;
  }

  // ---- gather geometry: item j of this thread is float4 #k4 of halo pixel p = (gtid>>2) + 64*j ----
  const int k4 = gtid & 3;
  int pix[APT];
#pragma unroll
  for (int j = 0; j < APT; ++j) {
    const int p = (gtid >> 2) + (GT / 4) * j;
    int v = -1;
    if (p < WG_NPIX) {
      const int hy = p / WG_PW;
      const int hx = p - hy * WG_PW;
      const int iy = y0 - 1 + hy;
      const int ix = x0 - 1 + hx;
      if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) v = (b * a.h + iy) * a.w + ix;
    }
    pix[j] = v;
  }
  // unconditional loads from clamped addresses (a load under a divergent branch is waited for on the spot); out-of-image
  // items are zeroed in store_patch -- touching the value here would make the wave wait for the fetch right away
  // The GroupNorm scale / shift rows of the slab travel with the activations (fetched a slab ahead): loading them inside
  // store_patch put an s_waitcnt vmcnt(0) there, which also drained the weight fragments requested just before.
  const bool affine = a.prologue == CF_PRO_AFFINE || a.prologue == CF_PRO_AFFINE_SWISH;
  // H2: range scale of an un-normalised input (cf_conv_desc.act_scale): powers of two, x * s and acc / s are exact; 1 when unused
  float act_s = 1.f, act_is = 1.f;
  if (H2 && !affine && a.act_scale) {
    act_s = a.act_scale[2 * b];
    act_is = a.act_scale[2 * b + 1];
  }
  const float act_s02 = 0.2f * act_s;  // LeakyReLU slope folded with the scale: fl(y * (0.2 s)) == fl(0.2 y) * s
  const float* const tab_sc = affine ? a.pro_scale + (size_t)b * a.cin : a.weight;  // (any valid address when unused)
  const float* const tab_sh = affine ? a.pro_shift + (size_t)b * a.cin : a.weight;
  f32x4 rsc, rsh;
  auto load_A = [&](int chunk, f32x4(&ra)[APT]) {
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
      ra[j] = *reinterpret_cast<const f32x4*>(src + (size_t)pj * cs + cc);
    }
  };
  // prologue + write to the LDS patch; zero padding stays exactly zero (it pads the post-activation tensor)
  auto store_patch_mode = [&](const f32x4(&ra)[APT], int chunk, auto mode) {
    constexpr int PRO = decltype(mode)::value;
    const f32x4 sc = rsc, sh = rsh;  // (fetched with the activations; only read by the affine modes)
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      const int p = (gtid >> 2) + (GT / 4) * j;  // (p >= 180: padding rows of the patch buffer, written as zeros -- no branch)
      {
        const bool valid = pix[j] >= 0;
        f32x4 v = ra[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y = v[e];
          if (PRO == CF_PRO_AFFINE) y = y * sc[e] + sh[e];
          if (PRO == CF_PRO_AFFINE_SWISH) {
            y = y * sc[e] + sh[e];
            y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));  // same hardware exp / rcp swish as the direct kernel
          }
          if (PRO == CF_PRO_LEAKY) y = H2 ? y * (y > 0.f ? act_s : act_s02) : (y > 0.f ? y : 0.2f * y);
          if (H2 && PRO == CF_PRO_NONE) y = y * act_s;
          v[e] = valid ? y : 0.f;
        }
        *reinterpret_cast<f32x4*>(patch + p * CF_LDK + k4 * 4) = v;
      }
    }
  };
  auto store_patch = [&](const f32x4(&ra)[APT], int chunk) {
    switch (a.prologue) {
      case CF_PRO_AFFINE: store_patch_mode(ra, chunk, std::integral_constant<int, CF_PRO_AFFINE>{}); break;
      case CF_PRO_AFFINE_SWISH: store_patch_mode(ra, chunk, std::integral_constant<int, CF_PRO_AFFINE_SWISH>{}); break;
      case CF_PRO_LEAKY: store_patch_mode(ra, chunk, std::integral_constant<int, CF_PRO_LEAKY>{}); break;
      default: store_patch_mode(ra, chunk, std::integral_constant<int, CF_PRO_NONE>{}); break;
    }
  };
  // input transform: item -> (tile, channel quad, xi row): 8 ds_read_b128, 8 vector adds, 4 ds_write_b128; 2 items per thread
  auto transform = [&]() {
#pragma unroll
    for (int it = gtid; it < 512; it += GT) {
      const int t_c4 = it & 3, t_xi = (it >> 2) & 3, t_tile = it >> 4;
      const int t_ty = t_tile >> 3, t_tx = t_tile & 7;  // tile = ty*8 + tx, outputs (2ty..2ty+1, 2tx..2tx+1) of the patch
      // B^T d along rows: xi0 = r0 - r2, xi1 = r1 + r2, xi2 = r2 - r1, xi3 = r1 - r3  ->  two patch rows per item, combined
      // as sa*ra + sb*rb with sa, sb = +-1 (exact): a per-lane choice of add / sub compiled into divergent branches with an
      // LDS wait inside each, which serialised the eight reads of an item
      const int ra_ = t_xi == 0 ? 0 : 1, rb_ = t_xi == 3 ? 3 : 2;
      const float sa = t_xi == 2 ? -1.f : 1.f, sb = (t_xi == 1 || t_xi == 2) ? 1.f : -1.f;
      const float* pa = patch + ((2 * t_ty + ra_) * WG_PW + 2 * t_tx) * CF_LDK + t_c4 * 4;
      const float* pb = patch + ((2 * t_ty + rb_) * WG_PW + 2 * t_tx) * CF_LDK + t_c4 * 4;
      f32x4 t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 da = *reinterpret_cast<const f32x4*>(pa + c * CF_LDK);
        const f32x4 db = *reinterpret_cast<const f32x4*>(pb + c * CF_LDK);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[c][e] = __fmaf_rn(db[e], sb, da[e] * sa);
      }
      // (.) B along columns: nu0 = t0 - t2, nu1 = t1 + t2, nu2 = t2 - t1, nu3 = t1 - t3
      if constexpr (H2) {
        // operand format, written ONCE here: a V row holds the slab's 16 channels as [hi: 16 halves | lo: 16 halves] (the 64 bytes the
        // fp32 values took); the conversion leaves the MMA stage's LDS-read -> MFMA chain for the transform phase, which the co-resident
        // workgroup's MFMAs overlap
        typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
        const f32x4 v4[4] = {v4sub(t[0], t[2]), v4add(t[1], t[2]), v4sub(t[2], t[1]), v4sub(t[1], t[3])};
        float* vo = V + t_xi * 4 * WG_PS + t_tile * CF_LDK + t_c4 * 2;  // position (xi, nu = 0): this item's two words of the hi half
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          float h0, l0, h1, l1;
          cf_split_pair(v4[nu][0], v4[nu][1], h0, l0);
          cf_split_pair(v4[nu][2], v4[nu][3], h1, l1);
          *reinterpret_cast<wg_f32x2*>(vo + nu * WG_PS) = wg_f32x2{h0, h1};
          *reinterpret_cast<wg_f32x2*>(vo + nu * WG_PS + 8) = wg_f32x2{l0, l1};
        }
      } else {
        float* vo = V + t_xi * 4 * WG_PS + t_tile * CF_LDK + t_c4 * 4;  // position (xi, nu = 0)
        *reinterpret_cast<f32x4*>(vo + 0 * WG_PS) = v4sub(t[0], t[2]);
        *reinterpret_cast<f32x4*>(vo + 1 * WG_PS) = v4add(t[1], t[2]);
        *reinterpret_cast<f32x4*>(vo + 2 * WG_PS) = v4sub(t[2], t[1]);
        *reinterpret_cast<f32x4*>(vo + 3 * WG_PS) = v4sub(t[1], t[3]);
      }
    }
  };

  f32x16 acc[4][NI];  // [nu][n tile]
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nu][ni][r] = 0.f;

  // this lane's B fragments.  The packed weights are stored in MFMA-operand order,
  //   U[pos = xi*4 + nu][chunk][n tile of 32][kg][lane][4]  (element = U[n = tile*32 + (lane & 31)][k = kg*8 + (lane >> 5)*4 + e]),
  // so every fragment load of a wave is one contiguous 1 KB block; they go global/L2 -> registers (each is used by one wave).
  const size_t pos_stride = (size_t)a.nchunks * a.cout_pad * CF_BK;
  const float* const wlane = a.weight + (size_t)(xi * 4) * pos_stride + (size_t)(n0 / 32) * 512 + lane * 4;
  const float* const alane = V + (xi * 4) * WG_PS + l31 * CF_LDK + half * 4;
  f32x4 bq[4][NI][2];
  auto load_B = [&](int chunk, int nu) {
    const float* wc = wlane + (size_t)chunk * a.cout_pad * CF_BK + nu * pos_stride;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) bq[nu][ni][kg] = *reinterpret_cast<const f32x4*>(wc + ni * 512 + kg * 256);
  };
  auto mma = [&](int nu) {
    if constexpr (H2) {
      // this lane's 8 channels of the slab (row = tile l31, channels half*8 .. +7), already in operand format (alane holds half*4)
      const f32x4 ah = *reinterpret_cast<const f32x4*>(alane + nu * WG_PS);      // 8 hi halves
      const f32x4 al = *reinterpret_cast<const f32x4*>(alane + nu * WG_PS + 8);  // 8 lo halves
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8, al), __builtin_bit_cast(wg_f16x8, bq[nu][ni][0]),
                                                             acc[nu][ni], 0, 0, 0);
        acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8, ah), __builtin_bit_cast(wg_f16x8, bq[nu][ni][1]),
                                                             acc[nu][ni], 0, 0, 0);
        acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8, ah), __builtin_bit_cast(wg_f16x8, bq[nu][ni][0]),
                                                             acc[nu][ni], 0, 0, 0);
      }
    } else {
      f32x4 aq[2];
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) aq[kg] = *reinterpret_cast<const f32x4*>(alane + nu * WG_PS + kg * 8);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[nu][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[kg][j], bq[nu][ni][kg][j], acc[nu][ni], 0, 0, 0);
    }
  };

  // ---- to the output domain: NI passes of 32 channels.  Every wave contracts its nu axis in registers and stages R[xi][bb] in the
  // (idle) V buffer; then item = (tile, output column bb, channel quad) contracts xi and owns two output pixels (rows aa = 0, 1):
  // o[it * 2 + aa] for the thread's two items.  Both barriers are inside, the leading one retires every wave's reads of V.
  float* const R = V;  // [(xi*2 + bb)][tile][WG_RLD]
  const int e_n4 = gtid & 7;  // (the item stride is a multiple of 8: both items of a thread have the same channel quad)
  auto to_output = [&](int pass, f32x4(&o)[4]) __attribute__((always_inline)) {
    __syncthreads();
    // staging addresses: one base per thread plus compile-time offsets (DS immediate fields) -- written with the row term inside the
    // index expression hipcc kept thirty-odd separate addresses alive (18 / 24 spilled registers in the split-K instantiations)
    const int wb = (xi * 2 * WG_NT + 4 * half) * WG_RLD + l31;   // R[xi][0][row = 4 half + ..][channel]
    const int rb = (((gtid >> 3) & 1) * WG_NT + (gtid >> 4)) * WG_RLD + e_n4 * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][pass][r], m1 = acc[1][pass][r], m2 = acc[2][pass][r], m3 = acc[3][pass][r];
      const int roff = ((r & 3) + 8 * (r >> 2)) * WG_RLD;  // (cf_acc_row without its lane term)
      R[wb + roff] = (m0 + m1) + m2;                    // nu axis: R[xi][0] = M0 + M1 + M2
      R[wb + WG_NT * WG_RLD + roff] = (m1 - m2) - m3;   //          R[xi][1] = M1 - M2 - M3
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // item (tile = gtid / 16 + 16 k, output column bb, channel quad)
      f32x4 x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(R + rb + (q * 2 * WG_NT + 16 * k) * WG_RLD);
      o[k * 2 + 0] = v4add(v4add(x[0], x[1]), x[2]);  // xi axis: Y[0][bb] = R0 + R1 + R2 ; Y[1][bb] = R1 - R2 - R3
      o[k * 2 + 1] = v4sub(v4sub(x[1], x[2]), x[3]);
    }
  };

  // SK: ordered sum of the finished virtual chunks, output domain: [NI*4][256 threads] float4 in LDS, private to each thread
  [[maybe_unused]] f32x4* const tot = reinterpret_cast<f32x4*>(V + WG_V_FLOATS) + gtid;
  if constexpr (SK) {
#pragma unroll
    for (int i = 0; i < NI * 4; ++i) tot[i * GT] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x4 ra[APT];
  load_A(kb, ra);
  for (int c = 0; c < n; ++c) {
    const int chunk = kb + c;
    // slot 0: gather-store.  Weight fragments of positions nu 0,1 are requested first: two slots of cover.
    load_B(chunk, 0);
    load_B(chunk, 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the fetches up here (hipcc would sink them next to their first use)
    store_patch(ra, chunk);
    __syncthreads();
    // slot 1: next slab's activations are requested (a whole slab of cover), then the transform
    load_A(c + 1 < n ? chunk + 1 : chunk, ra);  // unconditional (clamped): a load under a branch makes hipcc drain vmcnt
    __builtin_amdgcn_sched_barrier(0);
    transform();
    __builtin_amdgcn_sched_barrier(0);
    load_B(chunk, 2);  // the transform's registers are free again; consumed two barriers later
    load_B(chunk, 3);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // slot 2
    mma(0);
    mma(1);
    // slot 3
    mma(2);
    mma(3);
    // (the barrier after the next gather-store separates these reads of V from its rewrite)
    if constexpr (SK) {
      if ((c + 1) % CF_SK_SLABS == 0) {  // a virtual chunk is complete: fold its output-domain sum (one workgroup) or park it (split)
        f32x4 v[NI * 4];
#pragma unroll
        for (int pass = 0; pass < NI; ++pass) {
          f32x4 o[4];
          to_output(pass, o);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[pass * 4 + i] = o[i];
            tot[(pass * 4 + i) * GT] += o[i];
          }
        }
        if (a.nsplit > 1) cf_splitk_park(v, a.ws, sk_tile, chunk / CF_SK_SLABS, a.nchunks / CF_SK_SLABS, GT);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][ni][r] = 0.f;
        __syncthreads();  // staging (the V region) is rewritten by the next slab's transform
      }
    }
  }

  if constexpr (SK) {
    if (a.nsplit > 1) {
      f32x4 v[NI * 4];
      __syncthreads();
      if (!cf_splitk_finish(v, a.ws, a.counters, sk_tile, a.nchunks / CF_SK_SLABS, a.nsplit, GT, smem)) return;
#pragma unroll
      for (int i = 0; i < NI * 4; ++i) tot[i * GT] = v[i];
    }
  }

  // ---- epilogue: bias / residual / SFT, 16-byte stores, GroupNorm statistics of what was written -------------------------------
  // Residual / SFT operands of both 32-channel passes are requested first (the main loop's registers are free by now): their HBM
  // latency overlaps the LDS contraction instead of following it.  (n < cout always: the host requires cout % 64 == 0.)
  unsigned offs[NI][4];  // element offsets fit 32 bits (tensors < 16 GiB)
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
        if (a.epilogue == CF_EPI_RESIDUAL || a.epilogue == CF_EPI_SFT) r0[pass][k * 2 + aa] = *reinterpret_cast<const f32x4*>(a.res + off);
        if (a.epilogue == CF_EPI_SFT) r1[pass][k * 2 + aa] = *reinterpret_cast<const f32x4*>(a.sft_scale + off);
      }
    }
#pragma unroll
  for (int pass = 0; pass < NI; ++pass) {
    f32x4 o[4];
    if constexpr (SK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = tot[(pass * 4 + i) * GT];
    } else {
      to_output(pass, o);
    }
    const int nn = n0 + pass * 32 + e_n4 * 4;
    constexpr bool nvalid = true;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + nn);
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const float acc_s = a.acc_scale * act_is;  // (a product of powers of two: exact)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = o[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = H2 ? v[e] * acc_s + bias4[e] : v[e] + bias4[e];  // (a power of two: exact)
      if (a.epilogue == CF_EPI_RESIDUAL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r0[pass][i][e];
      } else if (a.epilogue == CF_EPI_SFT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r0[pass][i][e] + a.sft_w * (r0[pass][i][e] * r1[pass][i][e] + v[e]);
      }
      *reinterpret_cast<f32x4*>(a.out + offs[pass][i]) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ssum[e] += v[e];
        ssq[e] += v[e] * v[e];
      }
    }
    if (a.stats_out) {
      // GroupNorm statistics of the values just written (fp64 partials, fixed shuffle order): one partial per
      // (image, group, output patch, wave) -- nparts = tiles_per_img * 4
      const int cpg = a.stats_cpg;
      double d0, q0, d1 = 0, q1 = 0;
      if (cpg == 2) {
        d0 = (double)ssum[0] + ssum[1];
        q0 = (double)ssq[0] + ssq[1];
        d1 = (double)ssum[2] + ssum[3];
        q1 = (double)ssq[2] + ssq[3];
      } else {
        d0 = ((double)ssum[0] + ssum[1]) + ((double)ssum[2] + ssum[3]);
        q0 = ((double)ssq[0] + ssq[1]) + ((double)ssq[2] + ssq[3]);
      }
      for (int o2 = 8; o2 < 64; o2 <<= 1) {  // the (tile, bb) items of this wave: lanes with the same channel quad
        d0 += __shfl_xor(d0, o2, 64);
        q0 += __shfl_xor(q0, o2, 64);
      }
      if (cpg == 2) {  // (second group of the lane: 64-channel layers only -- a wave-uniform branch)
        for (int o2 = 8; o2 < 64; o2 <<= 1) {
          d1 += __shfl_xor(d1, o2, 64);
          q1 += __shfl_xor(q1, o2, 64);
        }
      }
      for (int o2 = 1; o2 * 4 < cpg; o2 <<= 1) {  // adjacent channel quads of one group (cpg >= 8)
        d0 += __shfl_xor(d0, o2, 64);
        q0 += __shfl_xor(q0, o2, 64);
      }
      if ((lane >> 3) == 0 && nvalid && (nn % cpg) == 0) {
        const size_t pidx = (size_t)rt * 4 + xi;
        const int ng = a.cout / cpg;
        double* op = a.stats_out + (((size_t)b * ng + nn / cpg) * a.nparts + pidx) * 2;
        op[0] = d0;
        op[1] = q0;
        if (cpg == 2) {
          op[(size_t)a.nparts * 2] = d1;
          op[(size_t)a.nparts * 2 + 1] = q1;
        }
      }
    }
  }
}

// U = G g G^T per (n, c), evaluated in fp64 and rounded once, stored in MFMA-operand order
// [pos][cin_pad/16][cout_pad/32][kg 2][lane 64][4]: element = U[n = tile*32 + (lane&31)][c = chunk*16 + kg*8 + (lane>>5)*4 + e]
__global__ void pack_weight_winograd_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, int nchunks,
                                            float* __restrict__ packed, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 3), ln = (int)((i >> 2) & 63), kg = (int)((i >> 8) & 1);
  long r = i >> 9;
  const int ntiles = cout_pad / 32;
  const int n = (int)(r % ntiles) * 32 + (ln & 31);
  r /= ntiles;
  const int chunk = (int)(r % nchunks);
  const int pos = (int)(r / nchunks);
  const int c = chunk * CF_BK + kg * 8 + (ln >> 5) * 4 + e;
  float val = 0.f;
  if (n < cout && c < cin) {
    const float* g = w + ((long)n * cin + c) * 9;
    const int xi = pos >> 2, nu = pos & 3;
    // row xi of G g : combination of the three kernel rows
    double row[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const double g0 = g[x], g1 = g[3 + x], g2 = g[6 + x];
      row[x] = xi == 0 ? g0 : (xi == 1 ? 0.5 * (g0 + g1 + g2) : (xi == 2 ? 0.5 * (g0 - g1 + g2) : g2));
    }
    const double u = nu == 0 ? row[0] : (nu == 1 ? 0.5 * (row[0] + row[1] + row[2]) : (nu == 2 ? 0.5 * (row[0] - row[1] + row[2]) : row[2]));
    val = (float)u;
  }
  packed[i] = val;
}

// H2 operands: U' = scale * G g G^T (fp64, rounded once to fp32) as hi = f16(U'), lo = f16(U' - hi), in MFMA-operand order
// [pos][cin_pad/16][cout_pad/32][part: hi, lo][lane 64][4 words]; a lane's 16 bytes are the 8 halves of
// U'[n = tile*32 + (lane&31)][c = chunk*16 + (lane>>5)*8 + 0..7]  (v_mfma_f32_32x32x16_f16 B operand).
// bf16_single: the hi slot holds bf16(U') instead and the lo slot zeros (the single-operand bf16 form of cf_wsplit.hip; its IEEE-half
// form reads the hi slot of the split packing as it is).
__global__ void pack_weight_winograd_f16x2_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, int nchunks, float scale,
                                                  unsigned* __restrict__ packed, long total, int bf16_single) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one 32-bit word = two halves
  if (i >= total) return;
  const int e = (int)(i & 3), ln = (int)((i >> 2) & 63), part = (int)((i >> 8) & 1);
  long r = i >> 9;
  const int ntiles = cout_pad / 32;
  const int n = (int)(r % ntiles) * 32 + (ln & 31);
  r /= ntiles;
  const int chunk = (int)(r % nchunks);
  const int pos = (int)(r / nchunks);
  const int xi = pos >> 2, nu = pos & 3;
  unsigned out = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = chunk * CF_BK + (ln >> 5) * 8 + e * 2 + h;
    float val = 0.f;
    if (n < cout && c < cin) {
      const float* g = w + ((long)n * cin + c) * 9;
      double row[3];
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const double g0 = g[x], g1 = g[3 + x], g2 = g[6 + x];
        row[x] = xi == 0 ? g0 : (xi == 1 ? 0.5 * (g0 + g1 + g2) : (xi == 2 ? 0.5 * (g0 - g1 + g2) : g2));
      }
      const double u = nu == 0 ? row[0] : (nu == 1 ? 0.5 * (row[0] + row[1] + row[2]) : (nu == 2 ? 0.5 * (row[0] - row[1] + row[2]) : row[2]));
      val = (float)(u * (double)scale);
    }
    if (bf16_single) {
      const __bf16 bv = (__bf16)(part ? 0.f : val);
      out |= (unsigned)__builtin_bit_cast(unsigned short, bv) << (16 * h);
    } else {
      const _Float16 hi = (_Float16)val;
      const _Float16 hv = part ? (_Float16)(val - (float)hi) : hi;
      out |= (unsigned)__builtin_bit_cast(unsigned short, hv) << (16 * h);
    }
  }
  packed[i] = out;
}

}  // namespace

static int pack_winograd_halves(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed, cf_stream_t stream,
                                int bf16_single) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_winograd_f16x2: null pointer");
  CF_REQUIRE(cin_pad % CF_BK == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 64 == 0,
             "cf_pack_conv_weight_winograd_f16x2: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  int ex = 0;
  CF_REQUIRE(scale > 0.f && frexpf(scale, &ex) == 0.5f, "cf_pack_conv_weight_winograd_f16x2: scale %g is not a power of two", (double)scale);
  const long total = 16L * cin_pad * cout_pad;  // 32-bit words: hi + lo half per weight
  hipLaunchKernelGGL(pack_weight_winograd_f16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     cout, cin, cout_pad, cin_pad / CF_BK, scale, reinterpret_cast<unsigned*>(packed), total, bf16_single);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_winograd_f16x2");
  return CF_OK;
}
extern "C" int cf_pack_conv_weight_winograd_f16x2(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed,
                                                  cf_stream_t stream) {
  return pack_winograd_halves(w, cout, cin, cout_pad, cin_pad, scale, packed, stream, 0);
}
extern "C" int cf_pack_conv_weight_winograd_bf16(const float* w, int cout, int cin, int cout_pad, int cin_pad, float scale, void* packed,
                                                 cf_stream_t stream) {
  return pack_winograd_halves(w, cout, cin, cout_pad, cin_pad, scale, packed, stream, 1);
}

extern "C" int cf_pack_conv_weight_winograd(const float* w, int cout, int cin, int cout_pad, int cin_pad, float* packed,
                                            cf_stream_t stream) {
  CF_REQUIRE(w && packed, "cf_pack_conv_weight_winograd: null pointer");
  CF_REQUIRE(cin_pad % CF_BK == 0 && cin_pad >= cin && cout_pad >= cout && cout_pad % 64 == 0,
             "cf_pack_conv_weight_winograd: bad padding cin %d->%d cout %d->%d", cin, cin_pad, cout, cout_pad);
  const long total = 16L * cin_pad * cout_pad;
  hipLaunchKernelGGL(pack_weight_winograd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     cout, cin, cout_pad, cin_pad / CF_BK, packed, total);
  CF_CHECK_LAUNCH("cf_pack_conv_weight_winograd");
  return CF_OK;
}

bool cf_wsplit_covers(const cf_conv_desc* d);                                        // cf_wsplit.hip: the eight-wave,
int cf_wsplit_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query);  // 128-channel split-half form

// Called by cf_conv2d (cf_igemm.hip) for descriptors with winograd != 0; the common argument checks have run there.
int cf_winograd_launch(const cf_conv_desc* d, hipStream_t stream, int* parts_query) {
  CF_REQUIRE(d->taps == 9 && d->stride == 1 && !d->upsample && !d->in_nchw && !d->out_nchw &&
                 d->bf16_mfma >= CF_OPERAND_F32 && d->bf16_mfma <= CF_OPERAND_F16X2,
             "cf_conv2d: winograd covers 3x3 stride-1 NHWC convolutions");
  const bool h1 = d->bf16_mfma == CF_OPERAND_F16 || d->bf16_mfma == CF_OPERAND_BF16;  // single 16-bit operands: the eight-wave kernel only
  CF_REQUIRE(!h1 || cf_wsplit_covers(d), "cf_conv2d(winograd, single 16-bit operands): needs cout %% 128 == 0 and at least 32x32 pixels per image");
  const bool h2 = d->bf16_mfma == CF_OPERAND_F16X2 || h1;
  CF_REQUIRE(!h2 || d->acc_scale > 0.f, "cf_conv2d(winograd, f16x2): acc_scale must be the inverse of the pack-time weight scale (got %g)",
             (double)d->acc_scale);
  CF_REQUIRE(d->hout % WG_TH == 0 && d->wout % WG_TW == 0, "cf_conv2d: winograd needs an output of %dx%d multiples (got %dx%d)",
             WG_TH, WG_TW, d->hout, d->wout);
  CF_REQUIRE(d->cout % 64 == 0 && d->cout_pad == d->cout, "cf_conv2d: winograd needs cout == cout_pad, a multiple of 64 (got %d / %d)", d->cout,
             d->cout_pad);
  CF_REQUIRE(d->epilogue == CF_EPI_NONE || d->epilogue == CF_EPI_RESIDUAL || d->epilogue == CF_EPI_SFT,
             "cf_conv2d: winograd epilogues are none / residual / SFT");
  CF_REQUIRE(d->pad_mode == CF_PAD_ZERO && (d->ld_in0 == 0 || d->ld_in0 == d->c0) && (d->ld_in1 == 0 || d->ld_in1 == d->c1) &&
                 (d->ld_out == 0 || d->ld_out == d->cout),
             "cf_conv2d: winograd reads / writes dense tensors with zero padding");
  if (h2 && cf_wsplit_covers(d)) return cf_wsplit_launch(d, stream, parts_query);
  CF_REQUIRE(!h1, "cf_conv2d(winograd, single 16-bit operands): not covered");
  WinoArgs a;
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
  a.acc_scale = h2 ? d->acc_scale : 1.f;
  a.act_scale = h2 ? d->act_scale : nullptr;
  a.out = d->out;
  a.stats_out = d->stats_out;
  a.stats_cpg = d->stats_cpg > 0 ? d->stats_cpg : 1;
  a.tiles_x = d->wout / WG_TW;
  a.tiles_per_img = a.tiles_x * (d->hout / WG_TH);
  a.nparts = a.tiles_per_img * 4;
  a.ntn = d->cout_pad / WG_BN;
  if (parts_query) {
    *parts_query = a.nparts;
    return CF_OK;
  }
  const bool sk = d->split_k >= 1;
  const size_t lds = (WG_PATCH_FLOATS + WG_V_FLOATS + (sk ? WG_NI * 4 * 256 * 4 : 0)) * sizeof(float);  // 61.7 KB (two workgroups per CU); SK: + 32 KB
  if (sk) {
    const int V = a.nchunks / CF_SK_SLABS;
    CF_REQUIRE(a.nchunks % CF_SK_SLABS == 0 && V >= 1 && V % d->split_k == 0,
               "cf_conv2d(winograd): split_k %d needs cin %% 128 == 0 and split_k dividing cin/128 = %d", d->split_k, V);
    CF_REQUIRE(d->split_k == 1 || (d->workspace && d->counters), "cf_conv2d(winograd): split_k > 1 needs workspace and counters");
  }
  // (cf_device_init sets the dynamic-LDS attribute of these four on each device)
  CF_LDS_ATTR((winograd_kernel<false, false>), (WG_PATCH_FLOATS + WG_V_FLOATS) * sizeof(float));
  CF_LDS_ATTR((winograd_kernel<false, true>), (WG_PATCH_FLOATS + WG_V_FLOATS) * sizeof(float));
  CF_LDS_ATTR((winograd_kernel<true, false>), (WG_PATCH_FLOATS + WG_V_FLOATS + WG_NI * 4 * 256 * 4) * sizeof(float));
  CF_LDS_ATTR((winograd_kernel<true, true>), (WG_PATCH_FLOATS + WG_V_FLOATS + WG_NI * 4 * 256 * 4) * sizeof(float));
  const int tiles = a.tiles_per_img * d->batch * a.ntn;
  if (sk) {
    WinoArgsSK k;
    static_cast<WinoArgs&>(k) = a;
    k.ws = d->workspace;
    k.counters = d->counters;
    k.nsplit = d->split_k;
    if (h2)
      hipLaunchKernelGGL((winograd_kernel<true, true>), dim3(tiles * d->split_k), dim3(256), lds, stream, k);
    else
      hipLaunchKernelGGL((winograd_kernel<true, false>), dim3(tiles * d->split_k), dim3(256), lds, stream, k);
  } else if (h2) {
    hipLaunchKernelGGL((winograd_kernel<false, true>), dim3(tiles), dim3(256), lds, stream, a);
  } else {
    hipLaunchKernelGGL((winograd_kernel<false, false>), dim3(tiles), dim3(256), lds, stream, a);
  }
  CF_CHECK_LAUNCH("cf_conv2d(winograd)");
  return CF_OK;
}

int cf_winograd_splitk_geometry(const cf_conv_desc* d, int* tiles, long* bytes_per_part) {
  CF_REQUIRE(d->hout % WG_TH == 0 && d->wout % WG_TW == 0 && d->cout_pad % WG_BN == 0 && (d->c0 + d->c1) % (16 * CF_SK_SLABS) == 0,
             "cf_conv2d(winograd): split_k needs whole patches and cin %% 128 == 0");
  *tiles = (d->hout / WG_TH) * (d->wout / WG_TW) * d->batch * (d->cout_pad / WG_BN);
  const int V = (d->c0 + d->c1) / (16 * CF_SK_SLABS);
  *bytes_per_part = (long)WG_NI * 4 * 256 * 16 * V / (d->split_k > 0 ? d->split_k : 1);  // output-domain chunk sums: 32 KB each
  return CF_OK;
}
