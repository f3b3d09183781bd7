// Shared helpers for libcodeformer_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "codeformer_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local last error (defined in cf_api.hip)
void cf_set_error(const char* fmt, ...);

#define CF_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      cf_set_error(__VA_ARGS__);     \
      return CF_ERR_ARG;             \
    }                                \
  } while (0)

#define CF_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      cf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return CF_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

// Kernels that launch with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, a PER-DEVICE property of the
// function.  The library keeps no mutable state: every such kernel enters a table while the shared object's static initialisers run
// (cf_lds_attr<kernel, bytes>::reg below, named by its launch site; the table is constant afterwards) and cf_device_init() -- called once
// per device by the host, codeformer_amd/lib.py -- sets the attribute of every entry on the current device.
int cf_register_kernel_lds(const void* kernel, int lds_bytes);  // cf_misc.hip
template <auto Kernel, int LdsBytes>
struct cf_lds_attr {
  static inline const int reg = cf_register_kernel_lds(reinterpret_cast<const void*>(Kernel), LdsBytes);
};
#define CF_LDS_ATTR(kernel, bytes) ((void)cf_lds_attr<kernel, (int)(bytes)>::reg)

// K-slab geometry shared by every MFMA kernel: a slab is 16 k-values wide; LDS rows are padded to
// 20 floats (80 B) so that 16 consecutive rows hit 16 distinct 16-byte bank slots under ds_read_b128.
constexpr int CF_BK = 16;
constexpr int CF_LDK = 20;

// One 16-wide K slab of  acc[mi][ni] += A[mi] * B[ni]^T  on v_mfma_f32_32x32x2_f32.
//   a_lds[mi] / b_lds[ni]: this lane's row in the A / B LDS slab, already offset by (lane>>5)*4 floats.
// Operand convention (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31].  Per 8-wide k group each lane reads ONE float4 of A and of B at k offset
// (l>>5)*4, so MFMA step j contracts k = {j, 4+j} of that group -- A and B use the same permutation
// of k, which leaves the dot product unchanged.
template <int MI, int NI>
__device__ __forceinline__ void cf_mma_slab(f32x16 (&acc)[MI][NI], const float* const (&a_lds)[MI],
                                            const float* const (&b_lds)[NI]) {
#pragma unroll
  for (int kg = 0; kg < CF_BK / 8; ++kg) {
    f32x4 af[MI], bf[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(a_lds[mi] + kg * 8);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(b_lds[ni] + kg * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
  }
}

// fp32 pair -> (hi, lo) IEEE-half pairs, x = hi + lo to 22 bits: one packed convert, then lo = f16(x - hi) as mixed-precision FMAs
// that read the f16 halves of `hi` directly and round once to f16 (x - hi is exact in fp32): 3 instructions per pair instead of 8
__device__ __forceinline__ void cf_split_pair(float x0, float x1, float& hi, float& lo) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t h = {(_Float16)x0, (_Float16)x1};
  hi = __builtin_bit_cast(float, h);
  float l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
  // `s_nop 1` INSIDE the string: hipcc pads nothing after inline asm, and a VGPR just written by a VALU instruction needs two wait states before
  // an MFMA may read it as an operand (cdna_hip_programming.md 5.7 item 2).  Where `lo` goes straight into an MFMA (the token GEMMs, the
  // four-wave Winograd kernel) the matrix instruction otherwise reads the stale register on some waves of some launches: found in round 5 on a
  // build whose schedule put the MFMA directly behind this statement (wrong, non-repeatable lo products: errors of 2e-4).
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(l) : "v"(hi), "v"(x1));
  lo = l;
}

// Output stores of the big epilogues: non-temporal when the launch's output tensor is larger than anything the caches keep for the
// next launch (CF_NT_STORE_BYTES; round 5: the first conv 0.42 -> 0.34 ms, the 64-channel 512x512 F(4,3) layers -1..3 %, 128-channel
// 128x128 -5 %: no write-allocate traffic through L2).  A host decision per launch (tensor bytes), never a change of the stored bits;
// one-face calls stay below the threshold, where the next kernel still finds its input in the Infinity Cache.
constexpr long CF_NT_STORE_BYTES = 100L << 20;
bool cf_nt_store(long out_bytes);   // cf_misc.hip: out_bytes >= the threshold (CF_NT_STORE_MB in the environment overrides it; 0 = never)
__device__ __forceinline__ void cf_store16(float* p, f32x4 v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}

// ---- bf16 STORAGE of activations (cf_conv_desc.io_bf16, ABI v22; precision 'bf16' of the network: BASELINE configs 3 / 5) ------------
// A tensor element is the upper half of its fp32 value, rounded to nearest even ONCE, in the producing epilogue, after the GroupNorm
// partials were taken from the fp32 values; consumers widen on load (a shift: exact).  Four channels = 8 bytes, eight = 16.
typedef unsigned cf_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned cf_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 cf_bf16x4_widen(cf_u32x2 p) {
  return f32x4{__builtin_bit_cast(float, p[0] << 16), __builtin_bit_cast(float, p[0] & 0xffff0000u), __builtin_bit_cast(float, p[1] << 16),
               __builtin_bit_cast(float, p[1] & 0xffff0000u)};
}
__device__ __forceinline__ cf_u32x2 cf_bf16x4_round(f32x4 v) {   // v_cvt_pk_bf16_f32 (round to nearest even) x 2
  const cf_bf16x2 a = {(__bf16)v[0], (__bf16)v[1]}, b = {(__bf16)v[2], (__bf16)v[3]};
  return cf_u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}
__device__ __forceinline__ f32x4 cf_load4_bf16(const float* base, size_t elem) {   // `base` points at bf16 elements (the descriptor keeps float* fields)
  return cf_bf16x4_widen(*reinterpret_cast<const cf_u32x2*>(reinterpret_cast<const unsigned short*>(base) + elem));
}
__device__ __forceinline__ void cf_store4_bf16(float* base, size_t elem, f32x4 v) {
  *reinterpret_cast<cf_u32x2*>(reinterpret_cast<unsigned short*>(base) + elem) = cf_bf16x4_round(v);
}

// Row of accumulator register r (0..15) of a 32x32 MFMA tile held by `lane`; the column is lane&31.
__device__ __forceinline__ int cf_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float cf_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float cf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- deterministic split-K across workgroups --------------------------------------------------------------------------------------
// A small-M layer (one face: 16x16 .. 64x64 pixels) has too few output tiles for 256 CUs, and its latency is the serial K loop of
// one workgroup.  The K range of such a layer is ALWAYS cut into V "virtual chunks" of CF_SK_SLABS slabs; each chunk is accumulated
// from zero and the chunk sums are added in chunk order starting from zero:   out = ((0 + P0) + P1) + ... + P(V-1).
// That association is fixed by the layer's shape, so the HOST may give the V chunks of a tile to 1, 2, .. V workgroups (by how many
// faces are in flight) without changing a single bit of the result:
//   nsplit == 1: one workgroup walks all chunks and folds each into a second accumulator in registers;
//   nsplit  > 1: every workgroup parks each of its chunk sums in a workspace (lane-contiguous 16-byte stores) and takes a ticket on the
//                tile's counter; the one drawing the last ticket adds all V chunk sums in order and runs the normal epilogue.
// Cross-workgroup visibility follows the agent-scope release / acquire recipe of cdna_hip_programming.md (Guideline 16): plain
// stores -> every wave drains vmcnt -> barrier -> one lane: release fence, drained, relaxed agent-scope ticket; the last arriver: one
// acquire fence -> barrier -> plain loads.  The counter is reset by the last arriver, so a zero-initialised counter buffer stays valid
// launch after launch.   ws layout: [tile][V][NV][nthreads] float4;  flag: one LDS dword inside the kernel's single LDS array.
constexpr int CF_TOKEN_IMAGE_MAX = 1024;  // 1x1 with split-half operands: larger images take the streaming convolution kernel, not the token GEMM
constexpr int CF_SK_SLABS = 8;  // 16-channel slabs per virtual chunk (128 K values)

template <int NV>
__device__ __forceinline__ void cf_splitk_park(const f32x4 (&v)[NV], float* __restrict__ ws, int tile, int vchunk, int V, int nthreads) {
  f32x4* mine = reinterpret_cast<f32x4*>(ws) + ((size_t)tile * V + vchunk) * NV * nthreads + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) mine[(size_t)i * nthreads] = v[i];
}

// returns true in the workgroup that drew the last ticket, with v = the ordered sum of all V chunk sums
template <int NV>
__device__ __forceinline__ bool cf_splitk_finish(f32x4 (&v)[NV], const float* __restrict__ ws, unsigned* __restrict__ counters, int tile, int V,
                                                 int nsplit, int nthreads, volatile float* lds_flag) {
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (restated where the compiler cannot drop it: the ticket must not overtake the write-back)
    const unsigned ticket = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<volatile unsigned*>(lds_flag) = ticket;
  }
  __syncthreads();
  const bool last = *reinterpret_cast<volatile unsigned*>(lds_flag) == (unsigned)(nsplit - 1);
  if (!last) return false;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    counters[tile] = 0;  // nobody else touches this tile's counter any more in this launch
  }
  __syncthreads();
  const f32x4* base = reinterpret_cast<const f32x4*>(ws) + (size_t)tile * V * NV * nthreads + tid;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < V; ++c) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += base[((size_t)c * NV + i) * nthreads];
  }
  return true;
}
