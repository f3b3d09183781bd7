// Softmax attention over 256 keys on fp32 MFMA for gfx950.
//
// Serves both attention flavours of the hot path (include/codeformer_hip.h, cf_attention):
//   * VQGAN AttnBlock: one head, head_dim 512, scale 512^-0.5 applied to q.k^T   (vqgan_arch.py:207-221)
//   * nn.MultiheadAttention core: 8 heads x 64, q pre-scaled by 1/8               (codeformer_arch.py:126)
//     (1/8 is a power of two, so scaling the scores instead of q is bit-identical).
//
// Workgroup = 256 threads, one (batch, head, 32-query tile).  Phase 1: S = Q K^T accumulated on
// v_mfma_f32_32x32x2_f32 with 16-wide k slabs of Q and K staged in LDS (each wave owns 64 keys).
// Phase 2: S*scale parked in LDS ([32][260] fp32), row softmax by wave-shuffle reductions (8 rows per wave,
// 4 keys per lane).  Phase 3: O = P V with P read from LDS as the A operand and 16-key slabs of V staged
// k-major in LDS as the B operand; output columns are processed 128 at a time (32 per wave).
#include "cf_common.h"


namespace {

constexpr int NKEY = 256;
constexpr int BQ = 32;
constexpr int PS = NKEY + 4;  // padded row stride of the score tile

// (The generic kernel of round 1 -- any head_dim, both operands through LDS slabs as described above -- served head_dim 64 until round 3 and
//  stayed as a timing fallback; it left the product in round 6: tools/experiments/ablation_and_timing_macros.patch.  The two kernels below
//  keep its phases and its arithmetic.)

// ---- head_dim 64 (the Transformer's nn.MultiheadAttention core) -------------------------------------------------------------------
// The generic kernel walks 4 + 16 barrier-separated slabs, each opening with a fetch it waits for, and half its waves idle in phase 3:
// 19 us per workgroup whatever the batch (21 us per launch for one face, nine launches per forward).  Here every operand is requested
// up front -- the four K slabs and the Q tile before phase 1, all of V (sixteen float4 per thread) before the softmax, which hides its
// latency -- the K slabs alternate between two LDS buffers (one barrier per slab), V is staged in two 128-key halves, and all four
// waves work in phase 3: wave w owns output columns 32 (w & 1) .. +31 and, of each half, keys 64 (w >> 1) .. +63; the two key groups
// meet through LDS.  Summation order of an output: keys [0, 64) + [128, 192) in one accumulator, [64, 128) + [192, 256) in the other,
// then their sum (fixed: results do not depend on batch or grid).
constexpr int A6_STAGE = (BQ + NKEY) * CF_LDK;                 // floats of one Q|K slab buffer
constexpr int A6_VLD = 72;                                     // V row stride: the two lane halves of a B read (4 keys apart) land 32 banks apart
constexpr int A6_VHALF = 128 * A6_VLD;                         // floats of a 128-key half of V
constexpr int A6_LDS_FLOATS = BQ * PS + (2 * A6_STAGE > A6_VHALF ? 2 * A6_STAGE : A6_VHALF);
constexpr size_t A6_LDS_BYTES = (size_t)A6_LDS_FLOATS * sizeof(float);

__global__ __launch_bounds__(256, 2) void attn64_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                        const float* __restrict__ V, int ldv, float* __restrict__ O, int ldo, float scale) {
  constexpr int DH = 64;
  extern __shared__ __attribute__((aligned(16))) float a6_smem[];
  float* const Ps = a6_smem;                 // [32][PS] scores, then probabilities
  float* const stage = a6_smem + BQ * PS;    // two Q|K slab buffers; later a V half; last the partial outputs of waves 2, 3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int q0 = blockIdx.x * BQ;
  const int h = blockIdx.y;
  const int b = blockIdx.z;
  const size_t rowbase = (size_t)b * NKEY;
  const int colbase = h * DH;

  // ---- all of K (four 16-channel slabs) and the Q tile: one exposed latency ----
  f32x4 kr[4][4];
  f32x4 qr[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = tid + 256 * j, row = f >> 2, k4 = f & 3;
      kr[sl][j] = *reinterpret_cast<const f32x4*>(K + (rowbase + row) * ldk + colbase + sl * CF_BK + k4 * 4);
    }
    const int row = (tid & 127) >> 2, k4 = tid & 3;  // (threads 128..255 fetch duplicates they never store: no divergent load)
    qr[sl] = *reinterpret_cast<const f32x4*>(Q + (rowbase + q0 + row) * ldq + colbase + sl * CF_BK + k4 * 4);
  }
  auto store_slab = [&](int sl, float* buf) __attribute__((always_inline)) {
    if (tid < BQ * 4) *reinterpret_cast<f32x4*>(buf + (tid >> 2) * CF_LDK + (tid & 3) * 4) = qr[sl];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = tid + 256 * j;
      *reinterpret_cast<f32x4*>(buf + (BQ + (f >> 2)) * CF_LDK + (f & 3) * 4) = kr[sl][j];
    }
  };
  // ---- phase 1: S[32][256] = Q K^T; wave w owns keys [64w, 64w + 64) ----
  f32x16 acc[1][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = acc[0][1][r] = 0.f;
  store_slab(0, stage);
  __syncthreads();
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    float* const buf = stage + (sl & 1) * A6_STAGE;
    if (sl + 1 < 4) store_slab(sl + 1, stage + ((sl + 1) & 1) * A6_STAGE);  // (its last readers left at the previous barrier)
    const float* ap[1] = {buf + l31 * CF_LDK + half * 4};
    const float* bp[2] = {buf + (BQ + wave * 64 + l31) * CF_LDK + half * 4, buf + (BQ + wave * 64 + 32 + l31) * CF_LDK + half * 4};
    cf_mma_slab<1, 2>(acc, ap, bp);
    __syncthreads();
  }
  // ---- all of V: 256 keys x 64 columns = sixteen float4 per thread, in flight across the score write-out and the softmax ----
  f32x4 vr[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int f = tid + 256 * j, key = f >> 4, c4 = f & 15;
    vr[j] = *reinterpret_cast<const f32x4*>(V + (rowbase + key) * ldv + colbase + c4 * 4);
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ps[cf_acc_row(r, lane) * PS + wave * 64 + ni * 32 + l31] = acc[0][ni][r] * scale;
  __syncthreads();
  // ---- phase 2: row softmax (F.softmax over keys), 8 rows per wave ----
#pragma unroll
  for (int i = 0; i < BQ / 4; ++i) {
    float* pr = Ps + (wave * (BQ / 4) + i) * PS + lane * 4;
    f32x4 v = *reinterpret_cast<f32x4*>(pr);
    const float m = cf_wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = expf(v[e] - m);
    const float sum = cf_wave_sum((v[0] + v[1]) + (v[2] + v[3]));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] / sum;
    *reinterpret_cast<f32x4*>(pr) = v;
  }
  // ---- phase 3: O[32][64] = P V ----
  const int nt = wave & 1, kgp = wave >> 1;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int vh = 0; vh < 2; ++vh) {
    if (vh) __syncthreads();  // every wave is done with the first half
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = tid + 256 * j;  // (key within the half, column quad) = (f >> 4, f & 15)
      *reinterpret_cast<f32x4*>(stage + (f >> 4) * A6_VLD + (f & 15) * 4) = vr[vh * 8 + j];
    }
    __syncthreads();  // (the first one also publishes the probabilities)
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
      const int kl = kgp * 64 + kg * 8 + half * 4;  // this lane half's four keys of the group (the k permutation of cf_mma_slab)
      const f32x4 af = *reinterpret_cast<const f32x4*>(Ps + l31 * PS + vh * 128 + kl);
#pragma unroll
      for (int j = 0; j < 4; ++j) o = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], stage[(kl + j) * A6_VLD + nt * 32 + l31], o, 0, 0, 0);
    }
  }
  __syncthreads();
  if (kgp == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[(nt * 32 + cf_acc_row(r, lane)) * 33 + l31] = o[r];
  }
  __syncthreads();
  if (kgp == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      O[(rowbase + q0 + cf_acc_row(r, lane)) * ldo + colbase + nt * 32 + l31] = o[r] + stage[(nt * 32 + cf_acc_row(r, lane)) * 33 + l31];
  }
}

// ---- head_dim 512 (the VQGAN AttnBlock) -------------------------------------------------------------------------------------------
// The generic kernel above spends a head_dim-512 call on 8 workgroups per image that walk 32 + 64 barrier-separated 16-wide slabs:
// 108 us whatever the batch.  Here a workgroup is 8 waves and owns one (image, 32-query tile, 256-column half of the output).
// Only the A operands are shared between waves (the Q tile in phase 1, the probabilities in phase 3): they live in LDS, the Q tile
// staged ONCE.  The B operands are private to a wave -- its 32 keys of K, its 32 output columns of V -- so each lane loads them
// straight from global memory in MFMA operand order, 64 channels / 64 keys ahead in registers: no barrier inside either loop.
// The products are summed in the order of the generic kernel, so the results are bit-identical to it.  The second column half
// recomputes S (fp32 MFMA time per workgroup: 6.8 us + 3.4 us).
constexpr int A5_THREADS = 512;
constexpr int A5_CW = 256;                        // output columns per workgroup
constexpr int A5_QS = (512 / CF_BK) * BQ * CF_LDK;  // floats: the whole Q tile as 32 sub-slabs [32 rows][16 + 4]
constexpr size_t A5_LDS_BYTES = (size_t)(BQ * PS + A5_QS) * sizeof(float);

__global__ __launch_bounds__(A5_THREADS) void attn512_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                             const float* __restrict__ V, int ldv, float* __restrict__ O, int ldo,
                                                             float scale) {
  constexpr int DH = 512;
  extern __shared__ __attribute__((aligned(16))) float a5_smem[];
  float* const Ps = a5_smem;            // [32][260] scores / probabilities
  float* const Qs = a5_smem + BQ * PS;  // [32 sub-slabs][32][20]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int q0 = blockIdx.x * BQ;
  const int h = blockIdx.y >> 1, chunk = blockIdx.y & 1;
  const int b = blockIdx.z;
  const size_t rowbase = (size_t)b * NKEY;
  const int colbase = h * DH;

  // this lane's K row (phase 1 B operand), first group of 64 channels already in flight while Q is staged
  const float* const krow = K + (rowbase + wave * 32 + l31) * ldk + colbase + half * 4;
  f32x4 kb[2][8];
  auto kload = [&](int g, f32x4(&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const f32x4*>(krow + g * 64 + i * 8);  // i = sub-slab * 2 + kg
  };
  kload(0, kb[0]);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = tid + A5_THREADS * j, row = f >> 7, k4 = f & 127;
    *reinterpret_cast<f32x4*>(Qs + ((k4 >> 2) * BQ + row) * CF_LDK + (k4 & 3) * 4) =
        *reinterpret_cast<const f32x4*>(Q + (rowbase + q0 + row) * ldq + colbase + k4 * 4);
  }
  __syncthreads();

  // ---------------- phase 1: S[32][256] = Q K^T ; wave w owns keys [32w, 32w+32) ----------------
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int g = 0; g < DH / 64; ++g) {
    if (g + 1 < DH / 64) kload(g + 1, kb[(g + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (the scheduler otherwise sinks every load to its use)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(Qs + ((g * 4 + (i >> 1)) * BQ + l31) * CF_LDK + (i & 1) * 8 + half * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], kb[g & 1][i][j], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) Ps[cf_acc_row(r, lane) * PS + wave * 32 + l31] = acc[r] * scale;

  // this lane's V column (phase 3 B operand): the first 64 keys on their way while the softmax runs
  const float* const vcol = V + (rowbase + half * 4) * ldv + colbase + chunk * A5_CW + wave * 32 + l31;
  float vb[2][32];
  auto vload = [&](int g, float(&dst)[32]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)  // i = 16-key slab * 2 + kg
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[i * 4 + j] = vcol[(size_t)(g * 64 + i * 8 + j) * ldv];
  };
  vload(0, vb[0]);
  __syncthreads();

  // ---------------- phase 2: row softmax (F.softmax over keys), 4 rows per wave ----------------
#pragma unroll
  for (int i = 0; i < BQ / 8; ++i) {
    float* pr = Ps + (wave * (BQ / 8) + i) * PS + lane * 4;
    f32x4 v = *reinterpret_cast<f32x4*>(pr);
    const float m = cf_wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = expf(v[e] - m);
    const float s = cf_wave_sum((v[0] + v[1]) + (v[2] + v[3]));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] / s;
    *reinterpret_cast<f32x4*>(pr) = v;
  }
  __syncthreads();

  // ---------------- phase 3: O[32][256 of 512] = P V, wave w owns columns [32w, 32w+32) of the chunk ----------------
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int g = 0; g < NKEY / 64; ++g) {
    if (g + 1 < NKEY / 64) vload(g + 1, vb[(g + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(Ps + l31 * PS + g * 64 + i * 8 + half * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) o = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], vb[g & 1][i * 4 + j], o, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    O[(rowbase + q0 + cf_acc_row(r, lane)) * ldo + colbase + chunk * A5_CW + wave * 32 + l31] = o[r];
}

}  // namespace

extern "C" int cf_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                            int batch, int heads, int head_dim, int nkeys, float scale, cf_stream_t stream) {
  CF_REQUIRE(q && k && v && out, "cf_attention: null pointer");
  CF_REQUIRE(nkeys == NKEY, "cf_attention: built for %d keys (got %d)", NKEY, nkeys);
  CF_REQUIRE(batch > 0 && heads > 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo >= heads * head_dim,
             "cf_attention: bad dims");
  const dim3 grid(NKEY / BQ, heads, batch), block(256);
  if (head_dim == 64) {
    CF_LDS_ATTR(attn64_kernel, A6_LDS_BYTES);  // (cf_device_init sets the dynamic-LDS attribute on each device)
    hipLaunchKernelGGL(attn64_kernel, grid, block, A6_LDS_BYTES, (hipStream_t)stream, q, ldq, k, ldk, v, ldv, out, ldo, scale);
  } else if (head_dim == 512) {
    CF_LDS_ATTR(attn512_kernel, A5_LDS_BYTES);
    hipLaunchKernelGGL(attn512_kernel, dim3(NKEY / BQ, heads * 2, batch), dim3(A5_THREADS), A5_LDS_BYTES, (hipStream_t)stream, q, ldq,
                       k, ldk, v, ldv, out, ldo, scale);
  } else {
    cf_set_error("cf_attention: head_dim %d unsupported (64 or 512)", head_dim);
    return CF_ERR_ARG;
  }
  CF_CHECK_LAUNCH("cf_attention");
  return CF_OK;
}
