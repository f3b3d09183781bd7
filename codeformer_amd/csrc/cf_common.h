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
