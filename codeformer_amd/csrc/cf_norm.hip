// GroupNorm statistics (two-stage, deterministic fp64 partials) and LayerNorm for gfx950.
// HBM-bound kernels: 16-byte coalesced loads along the channel axis of channels-last tensors.
#include "cf_common.h"

namespace {

// ---- GroupNorm stage 1 --------------------------------------------------------------------------
// x: [batch][hw][C] fp32.  grid (nblk, batch), 256 threads.  Thread t owns channel quad t % (C/4) and
// walks pixel rows  row0 + t/(C/4), += 1024/C.  Sums are carried in fp64 (the pass is HBM-bound; the fp64
// adds hide under the loads) so the later mean / E[x^2]-mean^2 is accurate for any |mean|/std.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int hw, int C, int cpg,
                                                       int rows_per_blk, double* __restrict__ part, int nblk) {
  __shared__ double red[256][4];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int cq4 = C >> 2;
  const int cq = tid % cq4;
  const int pr = tid / cq4;
  const int lanes = 256 / cq4;
  const int row0 = blk * rows_per_blk;
  const int row1 = min(hw, row0 + rows_per_blk);
  double s0 = 0, q0 = 0, s1 = 0, q1 = 0;
  const float* base = x + (size_t)b * hw * C + cq * 4;
  auto accum = [&](const f32x4& v) {
    const double a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
    s0 += a0 + a1;
    q0 += a0 * a0 + a1 * a1;
    s1 += a2 + a3;
    q1 += a2 * a2 + a3 * a3;
  };
  int r = row0 + pr;
  // 4 independent 16-byte loads in flight per thread (the pass is HBM-bound; one load per iteration starves it)
  for (; r + 3 * lanes < row1; r += 4 * lanes) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(base + (size_t)r * C);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(base + (size_t)(r + lanes) * C);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(base + (size_t)(r + 2 * lanes) * C);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(base + (size_t)(r + 3 * lanes) * C);
    accum(v0);
    accum(v1);
    accum(v2);
    accum(v3);
  }
  for (; r < row1; r += lanes) accum(*reinterpret_cast<const f32x4*>(base + (size_t)r * C));
  red[tid][0] = s0;
  red[tid][1] = q0;
  red[tid][2] = s1;
  red[tid][3] = q1;
  __syncthreads();
  const int G = C / cpg;
  if (tid < G) {
    double s = 0, q = 0;
    for (int t = 0; t < 256; ++t) {  // fixed order -> bitwise reproducible
      const int c = (t % cq4) * 4;
      if (c / cpg == tid) {
        s += red[t][0];
        q += red[t][1];
      }
      if ((c + 2) / cpg == tid) {
        s += red[t][2];
        q += red[t][3];
      }
    }
    double* o = part + (((size_t)b * G + tid) * nblk + blk) * 2;
    o[0] = s;
    o[1] = q;
  }
}

// ---- GroupNorm stage 2: fixed-order sum of partials -> per-(b,c) scale/shift --------------------
// One WORKGROUP per (image, normalisation group): the group's gmerge fine groups x nparts partials are contiguous in the
// table, so the 256 threads stream them with strided 16-byte loads on four independent chains each (conv-epilogue
// statistics can be thousands of partials per group; a single wave needed up to 60 us for them), reduce with a fixed
// shuffle butterfly + a fixed-order sum of the four waves (bitwise reproducible), and write the channel tables.  scale = gamma*rstd, shift = beta - mean*scale: the affine form ATen's CPU kernel applies.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, int batch, int nparts, int C,
                                                          int cpg, int gmerge, double inv_count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                          int ld) {
  __shared__ double red[4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int GM = C / (cpg * gmerge);
  const int b = blockIdx.x / GM, m = blockIdx.x - b * GM;
  const int G = C / cpg;
  const double2* p = reinterpret_cast<const double2*>(part) + ((size_t)b * G + (size_t)m * gmerge) * nparts;
  const int n = gmerge * nparts;
  // thread t sums partials t, t+256, ... on four independent chains (up to 4096 partials per group: the loads overlap)
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  int j = tid;
  for (; j + 768 < n; j += 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double2 v = p[j + 256 * u];
      s[u] += v.x;
      q[u] += v.y;
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (j + 256 * u < n) {
      const double2 v = p[j + 256 * u];
      s[u] += v.x;
      q[u] += v.y;
    }
  }
  double ss = (s[0] + s[1]) + (s[2] + s[3]), qq = (q[0] + q[1]) + (q[2] + q[3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o, 64);
    qq += __shfl_xor(qq, o, 64);
  }
  if (lane == 0) {
    red[wave][0] = ss;
    red[wave][1] = qq;
  }
  __syncthreads();
  ss = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  qq = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  const double mean = ss * inv_count;
  double var = qq * inv_count - mean * mean;
  if (var < 0) var = 0;
  const float fmean = (float)mean;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int cpm = cpg * gmerge;
  for (int i = tid; i < cpm; i += 256) {
    const int c = m * cpm + i;
    const float sc = rstd * gamma[c];
    scale[(size_t)b * ld + c] = sc;
    shift[(size_t)b * ld + c] = -sc * fmean + beta[c];
  }
}

// ---- per-image range scale for the 16-bit-operand convolutions (cf_conv_desc.act_scale) ----------------------------------------
// Two steps, both order-independent maxima (bitwise reproducible): ACT_CHUNKS workgroups per image reduce their share of the source
// -- MODE 0: sqrt of the largest statistics partial sumsq, a rigorous upper bound of max |x|; MODE 1: the exact maximum of the tensor --
// to one float each, then one wave per image turns the ACT_CHUNKS maxima into s = 2^k with growth * A * s in [2^13, 2^14).  (A single
// workgroup per image needed 37 us for the 1 MB of partials a 128-channel 256x256 layer writes per image.)
constexpr int ACT_CHUNKS = 32;

template <int MODE>
__global__ __launch_bounds__(256) void act_amax_kernel(const void* __restrict__ src, long n, float* __restrict__ chunk_max) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, ch = blockIdx.x;
  const long per = (n + ACT_CHUNKS - 1) / ACT_CHUNKS;   // elements (MODE 0: partials; MODE 1: float4 groups) of this chunk
  const long lo = ch * per, hi = lo + per < n ? lo + per : n;
  float m = 0.f;
  bool bad = false;
  if (MODE == 0) {
    const double2* p = reinterpret_cast<const double2*>(src) + (size_t)b * n;
    double q[4] = {0, 0, 0, 0};
    long j = lo + tid;
    for (; j + 768 < hi; j += 1024) {   // four independent loads in flight per thread
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double v = p[j + 256 * u].y;
        bad |= !(v == v) || v > 3.0e38;   // NaN / inf statistics: the image itself is not finite
        q[u] = v > q[u] ? v : q[u];
      }
    }
    for (; j < hi; j += 256) {
      const double v = p[j].y;
      bad |= !(v == v) || v > 3.0e38;
      q[0] = v > q[0] ? v : q[0];
    }
    const double qq = fmax(fmax(q[0], q[1]), fmax(q[2], q[3]));
    m = (float)sqrt(qq) * 1.0000002f;    // (rounded up: the bound must not fall below the true maximum)
  } else {
    const f32x4* p = reinterpret_cast<const f32x4*>(src) + (size_t)b * n;
    for (long j = lo + tid; j < hi; j += 256) {
      const f32x4 v = p[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bad |= !(v[e] == v[e]);
        m = fmaxf(m, fabsf(v[e]));
      }
    }
  }
  if (bad) m = __builtin_inff();
  m = cf_wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) chunk_max[b * ACT_CHUNKS + ch] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(64) void act_scale_finish_kernel(const float* __restrict__ chunk_max, float growth, float* __restrict__ act) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float m = lane < ACT_CHUNKS ? chunk_max[b * ACT_CHUNKS + lane] : 0.f;
  m = cf_wave_max(m) * growth;
  if (lane == 0) {
    int k = 0;
    if (m > 0.f && m < __builtin_inff()) {
      int e;
      (void)frexpf(m, &e);              // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(14 - e) in [2^13, 2^14)
      k = 14 - e;
      k = k < -100 ? -100 : (k > 100 ? 100 : k);
    }
    act[2 * b] = ldexpf(1.f, k);
    act[2 * b + 1] = ldexpf(1.f, -k);
  }
}

// ---- LayerNorm: one wave per row, C = 256*NV (NV float4 per lane) --------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int rows, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ pos, int npos, float* __restrict__ y,
                                                        float* __restrict__ ypos) {
  constexpr int C = NV * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = cf_wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q += d * d;
    }
  const float var = cf_wave_sum(q) * (1.0f / C);
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bt[e];
    *reinterpret_cast<f32x4*>(y + (size_t)row * C + c) = o;
    if (ypos) {
      const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (size_t)(row % npos) * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += pe[e];
      *reinterpret_cast<f32x4*>(ypos + (size_t)row * C + c) = o;
    }
  }
}

}  // namespace

extern "C" int cf_groupnorm_stats(const float* x, int batch, int hw, int c, int cpg, double* partial, int parts,
                                  cf_stream_t stream) {
  CF_REQUIRE(x && partial, "cf_groupnorm_stats: null pointer");
  CF_REQUIRE(c >= 16 && c <= 1024 && (1024 % c) == 0, "cf_groupnorm_stats: C=%d must divide 1024 and be >= 16", c);
  CF_REQUIRE(cpg >= 2 && (cpg % 2) == 0 && c % cpg == 0 && c / cpg <= 64, "cf_groupnorm_stats: bad cpg %d for C %d", cpg, c);
  CF_REQUIRE(parts >= 1 && batch >= 1 && hw >= 1, "cf_groupnorm_stats: bad dims");
  const int rows_per_blk = (hw + parts - 1) / parts;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(parts, batch), dim3(256), 0, (hipStream_t)stream, x, hw, c, cpg, rows_per_blk,
                     partial, parts);
  CF_CHECK_LAUNCH("cf_groupnorm_stats");
  return CF_OK;
}

// One launch: 32 workgroups per image reduce their slices of the statistics partials of up to TWO tensors (the halves of a concatenated
// input) -- or of a tensor itself -- and meet in two zero-initialised words per image: cells[2b] takes atomic maxima of the (non-negative)
// float bit patterns, cells[2b + 1] tickets; the workgroup drawing the last ticket turns the maximum into (s, 1 / s) and leaves both words
// at zero for the next launch.  The ticket is drawn only after the workgroup's own maximum has RETURNED from the memory-side atomic unit,
// so the last arriver's read sees every contribution without a cache write-back.
__global__ __launch_bounds__(256) void act_scale_fused_kernel(const double2* __restrict__ pa, long na, const double2* __restrict__ pb, long nb,
                                                              const f32x4* __restrict__ x, long nx, float growth, unsigned* __restrict__ cells,
                                                              float* __restrict__ act) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, ch = blockIdx.x;
  float m = 0.f;
  bool bad = false;
  double q[4] = {0, 0, 0, 0};
#pragma unroll
  for (int src = 0; src < 2; ++src) {
    const double2* p = src ? pb : pa;
    const long n = src ? nb : na;
    if (!p) continue;
    p += (size_t)b * n;
    const long per = (n + ACT_CHUNKS - 1) / ACT_CHUNKS;
    const long lo = ch * per, hi = lo + per < n ? lo + per : n;
    long j = lo + tid;
    for (; j + 768 < hi; j += 1024) {  // four independent loads in flight per thread
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double v = p[j + 256 * u].y;
        bad |= !(v == v) || v > 3.0e38;  // NaN / inf statistics: the image itself is not finite
        q[u] = v > q[u] ? v : q[u];
      }
    }
    for (; j < hi; j += 256) {
      const double v = p[j].y;
      bad |= !(v == v) || v > 3.0e38;
      q[0] = v > q[0] ? v : q[0];
    }
  }
  m = (float)sqrt(fmax(fmax(q[0], q[1]), fmax(q[2], q[3]))) * 1.0000002f;  // (rounded up: the bound must not fall below the true maximum)
  if (x) {
    const f32x4* p = x + (size_t)b * nx;
    const long per = (nx + ACT_CHUNKS - 1) / ACT_CHUNKS;
    const long lo = ch * per, hi = lo + per < nx ? lo + per : nx;
    for (long j = lo + tid; j < hi; j += 256) {
      const f32x4 v = p[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bad |= !(v[e] == v[e]);
        m = fmaxf(m, fabsf(v[e]));
      }
    }
  }
  if (bad) m = __builtin_inff();
  m = cf_wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const unsigned before = __hip_atomic_fetch_max(cells + 2 * b, __builtin_bit_cast(unsigned, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(before) : "memory");  // the maximum has been applied before the ticket is drawn
    const unsigned ticket = __hip_atomic_fetch_add(cells + 2 * b + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == ACT_CHUNKS - 1) {
      const unsigned bits = __hip_atomic_exchange(cells + 2 * b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cells + 2 * b + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float mm = __builtin_bit_cast(float, bits) * growth;
      int k = 0;
      if (mm > 0.f && mm < __builtin_inff()) {
        int e;
        (void)frexpf(mm, &e);  // mm = f * 2^e, f in [0.5, 1)  ->  mm * 2^(14 - e) in [2^13, 2^14)
        k = 14 - e;
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
      }
      act[2 * b] = ldexpf(1.f, k);
      act[2 * b + 1] = ldexpf(1.f, -k);
    }
  }
}

extern "C" int cf_act_scale_fused(const double* partial_a, int nper_a, const double* partial_b, int nper_b, const float* x, int64_t n_per_image,
                                  int batch, float growth, uint32_t* cells, float* act, cf_stream_t stream) {
  CF_REQUIRE(act && cells && batch >= 1 && growth > 0.f, "cf_act_scale_fused: bad arguments");
  CF_REQUIRE((partial_a != nullptr) != (x != nullptr), "cf_act_scale_fused: give statistics partials (one or two tensors) or a tensor, not both");
  CF_REQUIRE(!partial_a || nper_a >= 1, "cf_act_scale_fused: nper_a");
  CF_REQUIRE(!partial_b || (partial_a && nper_b >= 1), "cf_act_scale_fused: the second statistics array needs the first (and nper_b >= 1)");
  CF_REQUIRE(!x || (n_per_image >= 4 && n_per_image % 4 == 0), "cf_act_scale_fused: n_per_image %lld must be a positive multiple of 4", (long long)n_per_image);
  hipLaunchKernelGGL(act_scale_fused_kernel, dim3(ACT_CHUNKS, batch), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const double2*>(partial_a),
                     (long)nper_a, reinterpret_cast<const double2*>(partial_b), (long)nper_b, reinterpret_cast<const f32x4*>(x), (long)(n_per_image / 4),
                     growth, cells, act);
  CF_CHECK_LAUNCH("cf_act_scale_fused");
  return CF_OK;
}

extern "C" int cf_act_scale_from_stats(const double* partial, int batch, int nper, float growth, float* scratch, float* act, cf_stream_t stream) {
  CF_REQUIRE(partial && act && scratch && batch >= 1 && nper >= 1 && growth > 0.f, "cf_act_scale_from_stats: bad arguments");
  hipLaunchKernelGGL(act_amax_kernel<0>, dim3(ACT_CHUNKS, batch), dim3(256), 0, (hipStream_t)stream, (const void*)partial, (long)nper, scratch);
  hipLaunchKernelGGL(act_scale_finish_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, (const float*)scratch, growth, act);
  CF_CHECK_LAUNCH("cf_act_scale_from_stats");
  return CF_OK;
}

extern "C" int cf_act_scale_from_tensor(const float* x, int batch, int64_t n_per_image, float growth, float* scratch, float* act, cf_stream_t stream) {
  CF_REQUIRE(x && act && scratch && batch >= 1 && n_per_image >= 1 && growth > 0.f, "cf_act_scale_from_tensor: bad arguments");
  CF_REQUIRE(n_per_image % 4 == 0, "cf_act_scale_from_tensor: n_per_image %lld must be a multiple of 4 (16-byte loads)", (long long)n_per_image);
  hipLaunchKernelGGL(act_amax_kernel<1>, dim3(ACT_CHUNKS, batch), dim3(256), 0, (hipStream_t)stream, (const void*)x, (long)(n_per_image / 4), scratch);
  hipLaunchKernelGGL(act_scale_finish_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, (const float*)scratch, growth, act);
  CF_CHECK_LAUNCH("cf_act_scale_from_tensor");
  return CF_OK;
}

// ---- finalize of up to TWO tensors' partials (the halves of a concatenated GroupNorm input, codeformer_arch.py:152) and, optionally, the
// range-scale table of the same tensor(s) in ONE launch (round 6).  A workgroup owns one (image, merged group) of tensor a or b and runs
// gn_finalize_kernel's arithmetic on it -- the same loads in the same order, the same fixed-order sums: bitwise the tables of two
// cf_groupnorm_finalize launches.  While it walks its partials it also keeps their largest sum of squares; with `act` set the workgroups
// of an image meet in cells[2b] (atomic maximum of the non-negative float bits) / cells[2b + 1] (tickets) exactly as the workgroups of
// act_scale_fused_kernel do, and the last arriver writes (s, 1 / s): maxima are order-independent and sqrt / rounding are monotone, so
// the table is bitwise cf_act_scale_fused's.  One-face calls: 11 launches fewer per forward.
__global__ __launch_bounds__(256) void gn_finalize2_kernel(const double* __restrict__ part_a, int nparts_a, int c_a, int cpg_a, int gmerge_a,
                                                           const double* __restrict__ part_b, int nparts_b, int c_b, int cpg_b, int gmerge_b,
                                                           double inv_count, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift, int ld, float growth,
                                                           unsigned* __restrict__ cells, float* __restrict__ act) {
  __shared__ double red[4][2];
  __shared__ float redm[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int GMa = c_a / (cpg_a * gmerge_a), GMb = part_b ? c_b / (cpg_b * gmerge_b) : 0;
  const int per_img = GMa + GMb;
  const int b = blockIdx.x / per_img;
  int m = blockIdx.x - b * per_img;
  const bool second = m >= GMa;
  if (second) m -= GMa;
  const int nparts = second ? nparts_b : nparts_a, C = second ? c_b : c_a, cpg = second ? cpg_b : cpg_a, gmerge = second ? gmerge_b : gmerge_a;
  const int coff = second ? c_a : 0;
  const int G = C / cpg;
  const double2* p = reinterpret_cast<const double2*>(second ? part_b : part_a) + ((size_t)b * G + (size_t)m * gmerge) * nparts;
  const int n = gmerge * nparts;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0}, qm = 0;
  bool bad = false;
  int j = tid;
  for (; j + 768 < n; j += 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double2 v = p[j + 256 * u];
      s[u] += v.x;
      q[u] += v.y;
      bad |= !(v.y == v.y) || v.y > 3.0e38;
      qm = v.y > qm ? v.y : qm;
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (j + 256 * u < n) {
      const double2 v = p[j + 256 * u];
      s[u] += v.x;
      q[u] += v.y;
      bad |= !(v.y == v.y) || v.y > 3.0e38;
      qm = v.y > qm ? v.y : qm;
    }
  }
  double ss = (s[0] + s[1]) + (s[2] + s[3]), qq = (q[0] + q[1]) + (q[2] + q[3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o, 64);
    qq += __shfl_xor(qq, o, 64);
  }
  float mx = (float)sqrt(qm) * 1.0000002f;   // (act_scale_fused_kernel's bound of this slice)
  if (bad) mx = __builtin_inff();
  mx = cf_wave_max(mx);
  if (lane == 0) {
    red[wave][0] = ss;
    red[wave][1] = qq;
    redm[wave] = mx;
  }
  __syncthreads();
  ss = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  qq = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  const double mean = ss * inv_count;
  double var = qq * inv_count - mean * mean;
  if (var < 0) var = 0;
  const float fmean = (float)mean;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int cpm = cpg * gmerge;
  for (int i = tid; i < cpm; i += 256) {
    const int c = coff + m * cpm + i;
    const float sc = rstd * gamma[c];
    scale[(size_t)b * ld + c] = sc;
    shift[(size_t)b * ld + c] = -sc * fmean + beta[c];
  }
  if (act && tid == 0) {
    const float mm0 = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    const unsigned before = __hip_atomic_fetch_max(cells + 2 * b, __builtin_bit_cast(unsigned, mm0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(before) : "memory");  // the maximum has been applied before the ticket is drawn
    const unsigned ticket = __hip_atomic_fetch_add(cells + 2 * b + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (unsigned)per_img - 1u) {
      const unsigned bits = __hip_atomic_exchange(cells + 2 * b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cells + 2 * b + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float mm = __builtin_bit_cast(float, bits) * growth;
      int k = 0;
      if (mm > 0.f && mm < __builtin_inff()) {
        int e;
        (void)frexpf(mm, &e);
        k = 14 - e;
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
      }
      act[2 * b] = ldexpf(1.f, k);
      act[2 * b + 1] = ldexpf(1.f, -k);
    }
  }
}

extern "C" int cf_groupnorm_finalize2(const double* partial_a, int parts_a, int c_a, int cpg_a, int gmerge_a, const double* partial_b, int parts_b,
                                      int c_b, int cpg_b, int gmerge_b, int batch, int64_t count, const float* gamma, const float* beta, float eps,
                                      float* scale, float* shift, int ld, float act_growth, uint32_t* cells, float* act, cf_stream_t stream) {
  CF_REQUIRE(partial_a && gamma && beta && scale && shift && batch >= 1, "cf_groupnorm_finalize2: null pointer");
  CF_REQUIRE(cpg_a >= 1 && c_a % cpg_a == 0 && gmerge_a >= 1 && (c_a / cpg_a) % gmerge_a == 0 && parts_a >= 1 && count > 0,
             "cf_groupnorm_finalize2: bad dims of the first tensor (c=%d cpg=%d gmerge=%d parts=%d)", c_a, cpg_a, gmerge_a, parts_a);
  CF_REQUIRE(!partial_b || (cpg_b >= 1 && c_b % cpg_b == 0 && gmerge_b >= 1 && (c_b / cpg_b) % gmerge_b == 0 && parts_b >= 1 && cpg_a * gmerge_a == cpg_b * gmerge_b),
             "cf_groupnorm_finalize2: bad dims of the second tensor (c=%d cpg=%d gmerge=%d parts=%d; both halves share the output group size)", c_b, cpg_b, gmerge_b, parts_b);
  CF_REQUIRE(ld >= c_a + (partial_b ? c_b : 0), "cf_groupnorm_finalize2: ld %d smaller than the channel count", ld);
  CF_REQUIRE(!act || (cells && act_growth > 0.f), "cf_groupnorm_finalize2: the range-scale table needs the cells and a positive growth");
  const int per_img = c_a / (cpg_a * gmerge_a) + (partial_b ? c_b / (cpg_b * gmerge_b) : 0);
  hipLaunchKernelGGL(gn_finalize2_kernel, dim3((unsigned)(batch * per_img)), dim3(256), 0, (hipStream_t)stream, partial_a, parts_a, c_a, cpg_a, gmerge_a,
                     partial_b, parts_b, c_b, cpg_b, gmerge_b, 1.0 / (double)count, gamma, beta, eps, scale, shift, ld, act_growth, cells, act);
  CF_CHECK_LAUNCH("cf_groupnorm_finalize2");
  return CF_OK;
}

extern "C" int cf_groupnorm_finalize(const double* partial, int batch, int parts, int c, int cpg, int gmerge,
                                     int64_t count, const float* gamma, const float* beta, float eps, float* scale,
                                     float* shift, int ld, cf_stream_t stream) {
  CF_REQUIRE(partial && gamma && beta && scale && shift, "cf_groupnorm_finalize: null pointer");
  CF_REQUIRE(cpg >= 1 && c % cpg == 0 && gmerge >= 1 && (c / cpg) % gmerge == 0 && count > 0 &&
                 parts >= 1 && ld >= c,
             "cf_groupnorm_finalize: bad dims (c=%d cpg=%d gmerge=%d parts=%d ld=%d)", c, cpg, gmerge, parts, ld);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(batch * (c / (cpg * gmerge))), dim3(256), 0, (hipStream_t)stream, partial, batch, parts, c,
                     cpg, gmerge, 1.0 / (double)count, gamma, beta, eps, scale, shift, ld);
  CF_CHECK_LAUNCH("cf_groupnorm_finalize");
  return CF_OK;
}

extern "C" int cf_layernorm(const float* x, int rows, int c, const float* gamma, const float* beta, float eps,
                            const float* pos, int npos, float* y, float* ypos, cf_stream_t stream) {
  CF_REQUIRE(x && gamma && beta && y, "cf_layernorm: null pointer");
  CF_REQUIRE(!ypos || (pos && npos > 0), "cf_layernorm: ypos needs pos/npos");
  CF_REQUIRE(rows > 0, "cf_layernorm: rows");
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (c) {
    case 256: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, rows, gamma, beta, eps, pos, npos, y, ypos); break;
    case 512: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, rows, gamma, beta, eps, pos, npos, y, ypos); break;
    case 768: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, x, rows, gamma, beta, eps, pos, npos, y, ypos); break;
    case 1024: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, rows, gamma, beta, eps, pos, npos, y, ypos); break;
    default:
      cf_set_error("cf_layernorm: C=%d unsupported (256/512/768/1024)", c);
      return CF_ERR_ARG;
  }
  CF_CHECK_LAUNCH("cf_layernorm");
  return CF_OK;
}
