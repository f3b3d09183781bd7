// Small HBM-/latency-bound kernels of the CodeFormer path for gfx950: code argmax, codebook gather + AdaIN,
// nearest-code argmin, layout converters, the u8<->tensor boundary, and the two bundled StyleGAN2 ops.
#include <stdarg.h>

#include <cstdlib>

#include "cf_common.h"

// ------------------------------------------------------------------------------------------------
// library plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void cf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#ifndef CF_BUILD_ID
#define CF_BUILD_ID "unknown"
#endif
extern "C" int cf_version(void) { return CF_ABI_VERSION; }
static const char g_build_id[] = "CF_BUILD_ID=" CF_BUILD_ID;  // (the marker lets build.py read the id from the file without dlopen)
extern "C" const char* cf_build_id(void) { return g_build_id + 12; }
extern "C" const char* cf_last_error(void) { return g_err; }
// Table of the kernels that need the dynamic-LDS attribute (cf_common.h): filled by static initialisers while the library loads, constant
// afterwards.  Zero-initialised storage, so the order of the initialisers across translation units does not matter.
constexpr int CF_LDS_TABLE = 512;
static const void* g_lds_kernel[CF_LDS_TABLE];
static int g_lds_bytes[CF_LDS_TABLE];
static int g_lds_count;
int cf_register_kernel_lds(const void* kernel, int lds_bytes) {
  if (g_lds_count < CF_LDS_TABLE) {
    g_lds_kernel[g_lds_count] = kernel;
    g_lds_bytes[g_lds_count] = lds_bytes;
  }
  return g_lds_count++;
}
extern "C" int cf_device_init(void) {
  CF_REQUIRE(g_lds_count <= CF_LDS_TABLE, "cf_device_init: kernel table overflow (%d entries)", g_lds_count);
  for (int i = 0; i < g_lds_count; ++i) {
    const hipError_t e = hipFuncSetAttribute(g_lds_kernel[i], hipFuncAttributeMaxDynamicSharedMemorySize, g_lds_bytes[i]);
    if (e != hipSuccess) {
      cf_set_error("cf_device_init: hipFuncSetAttribute(%d B LDS) on kernel %d: %s", g_lds_bytes[i], i, hipGetErrorString(e));
      return CF_ERR_LAUNCH;
    }
  }
  return CF_OK;
}
bool cf_nt_store(long out_bytes) {
  static const long threshold = [] {
    const char* e = getenv("CF_NT_STORE_MB");
    const long mb = e ? atol(e) : -1;
    return mb < 0 ? CF_NT_STORE_BYTES : (mb == 0 ? (1L << 62) : mb << 20);
  }();
  return out_bytes >= threshold;
}
extern "C" int cf_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

namespace {

// ---- argmax over a row: one wave per row, wavefront-shuffle reduction, lowest index wins ties ----
__device__ __forceinline__ void argbest(float& v, int& i, float ov, int oi, bool want_max) {
  const bool better = want_max ? (ov > v) : (ov < v);
  if (better || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int rows, int n,
                                                          int64_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * n;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane * 4; c < n; c += 256) {  // ascending index within a lane
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) argbest(best, bi, v[e], c + e, true);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    argbest(best, bi, ov, oi, true);
  }
  if (lane == 0) idx[row] = bi;
}

// ---- nearest code: d_j = (zz + ee_j) - 2*s_j in the reference's operation order, argmin ----------
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ scores, const float* __restrict__ zz,
                                                        const float* __restrict__ ee, int rows, int n,
                                                        int64_t* __restrict__ idx, float* __restrict__ dmin) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = scores + (size_t)row * n;
  const float z2 = zz[row];
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane * 4; c < n; c += 256) {
    const f32x4 s = *reinterpret_cast<const f32x4*>(sr + c);
    const f32x4 e = *reinterpret_cast<const f32x4*>(ee + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) argbest(best, bi, (z2 + e[k]) - 2.0f * s[k], c + k, false);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    argbest(best, bi, ov, oi, false);
  }
  if (lane == 0) {
    idx[row] = bi;
    if (dmin) dmin[row] = best;
  }
}

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int rows, int dim,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane * 4; c < dim; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * dim + c);
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  s = cf_wave_sum(s);
  if (lane == 0) out[row] = s;
}

// ---- codebook gather (+AdaIN): one workgroup per (image, 32 channels); thread = (token lane 0..7, channel) ---------------------
// Statistics in fp64: every thread sums the tokens p = lane, lane + 8, ...; the eight partial sums of a channel are added in lane
// order through LDS (fixed order -> bitwise reproducible).  Unbiased variance as torch.var (codeformer_arch.py:23-25).
__global__ __launch_bounds__(256) void gather_adain_kernel(const int64_t* __restrict__ idx, const float* __restrict__ cb,
                                                           int ncodes, const float* __restrict__ lq, int ntok, int dim,
                                                           int adain, float eps, float* __restrict__ out) {
  extern __shared__ int s_idx[];                      // [ntok] clamped code indices, then 4 x [8][32] doubles
  double* red = reinterpret_cast<double*>(s_idx + ((ntok + 1) & ~1));
  __shared__ float s_stat[4][32];
  const int groups = (dim + 31) / 32;
  const int b = blockIdx.x / groups;
  const int c = (blockIdx.x - b * groups) * 32 + (threadIdx.x & 31);
  const int tl = threadIdx.x >> 5;
  const bool cv = c < dim;
  for (int p = threadIdx.x; p < ntok; p += blockDim.x) {
    long v = idx[(size_t)b * ntok + p];
    s_idx[p] = (int)(v < 0 ? 0 : (v >= ncodes ? ncodes - 1 : v));
  }
  __syncthreads();
  if (adain) {
    double s1 = 0, q1 = 0, s2 = 0, q2 = 0;
    if (cv)
      for (int p = tl; p < ntok; p += 8) {
        const double v1 = cb[(size_t)s_idx[p] * dim + c];
        const double v2 = lq[((size_t)b * ntok + p) * dim + c];
        s1 += v1;
        q1 += v1 * v1;
        s2 += v2;
        q2 += v2 * v2;
      }
    red[(0 * 8 + tl) * 32 + (threadIdx.x & 31)] = s1;
    red[(1 * 8 + tl) * 32 + (threadIdx.x & 31)] = q1;
    red[(2 * 8 + tl) * 32 + (threadIdx.x & 31)] = s2;
    red[(3 * 8 + tl) * 32 + (threadIdx.x & 31)] = q2;
    __syncthreads();
    if (tl == 0) {
      double t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t[k] = 0;
        for (int j = 0; j < 8; ++j) t[k] += red[(k * 8 + j) * 32 + threadIdx.x];
      }
      const double m1 = t[0] / ntok, m2 = t[2] / ntok;
      double var1 = (t[1] - t[0] * m1) / (ntok - 1), var2 = (t[3] - t[2] * m2) / (ntok - 1);  // unbiased (torch.var)
      if (var1 < 0) var1 = 0;
      if (var2 < 0) var2 = 0;
      s_stat[0][threadIdx.x] = (float)m1;
      s_stat[1][threadIdx.x] = sqrtf((float)var1 + eps);
      s_stat[2][threadIdx.x] = (float)m2;
      s_stat[3][threadIdx.x] = sqrtf((float)var2 + eps);
    }
    __syncthreads();
  }
  if (!cv) return;
  const float cm = adain ? s_stat[0][threadIdx.x & 31] : 0.f, cs = adain ? s_stat[1][threadIdx.x & 31] : 1.f;
  const float sm = adain ? s_stat[2][threadIdx.x & 31] : 0.f, ss = adain ? s_stat[3][threadIdx.x & 31] : 1.f;
  for (int p = tl; p < ntok; p += 8) {
    float v = cb[(size_t)s_idx[p] * dim + c];
    if (adain) v = ((v - cm) / cs) * ss + sm;  // codeformer_arch.py:42-43 operation order
    out[((size_t)b * ntok + p) * dim + c] = v;
  }
}

// ---- batched 2-D transpose [B][R][C] -> [B][C][R] through a padded LDS tile ---------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int R, int C, float* __restrict__ y) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (size_t)b * R * C;
  float* yb = y + (size_t)b * R * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < R && c < C) t[ty + 8 * i][tx] = xb[(size_t)r * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) yb[(size_t)c * R + r] = t[tx][ty + 8 * i];
  }
}

// ---- tensor boundary ------------------------------------------------------------------------------
// u8 HWC BGR -> fp32 NCHW RGB in [-1,1]: (float)((double)u / 255.) then (v - 0.5f) / 0.5f, the arithmetic of
// img2tensor(img / 255.) + normalize (inference_codeformer.py:199-201).  Only 256 inputs exist, so every workgroup evaluates them
// once into an LDS table; a thread then converts FOUR pixels: three 4-byte loads of the interleaved bytes, one 16-byte store per
// colour plane.  (hw % 4 != 0: the tail pixels of an image go one at a time.)
__global__ __launch_bounds__(256) void img_u8_to_tensor_kernel(const uint8_t* __restrict__ img, int hw, float* __restrict__ out,
                                                               long nquads, int batch) {
  __shared__ float lut[256];
  {
    const float v = (float)((double)threadIdx.x / 255.0);
    lut[threadIdx.x] = (v - 0.5f) / 0.5f;
  }
  __syncthreads();
  const int qpi = (hw + 3) >> 2;  // quads per image (the last one may be partial)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (long)gridDim.x * blockDim.x) {
    const long b = i / qpi;
    const int p = (int)(i - b * qpi) * 4;
    const uint8_t* px = img + (b * hw + p) * 3;
    float* o = out + b * 3 * hw + p;
    if (p + 4 <= hw && (hw & 3) == 0) {
      const uint32_t* w = reinterpret_cast<const uint32_t*>(px);  // 12-byte groups of a 4-byte aligned buffer: aligned
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];            // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
      const f32x4 r = {lut[(w0 >> 16) & 255], lut[(w1 >> 8) & 255], lut[w2 & 255], lut[w2 >> 24]};
      const f32x4 g = {lut[(w0 >> 8) & 255], lut[w1 & 255], lut[w1 >> 24], lut[(w2 >> 16) & 255]};
      const f32x4 bl = {lut[w0 & 255], lut[w0 >> 24], lut[(w1 >> 16) & 255], lut[(w2 >> 8) & 255]};
      *reinterpret_cast<f32x4*>(o) = r;
      *reinterpret_cast<f32x4*>(o + hw) = g;
      *reinterpret_cast<f32x4*>(o + 2 * (long)hw) = bl;
    } else {
      for (int k = 0; k < 4 && p + k < hw; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(long)c * hw + k] = lut[px[k * 3 + 2 - c]];
    }
  }
}

// fp32 NCHW RGB -> u8 HWC BGR == tensor2img(min_max=(-1,1)): clamp, (v + 1) / 2, * 255, round half to even (img_util.py:66-90);
// four pixels per thread: one 16-byte load per colour plane, three 4-byte stores of the interleaved bytes.
__device__ __forceinline__ uint32_t cf_to_u8(float v) {
  v = fminf(fmaxf(v, -1.0f), 1.0f);
  v = (v - (-1.0f)) / (1.0f - (-1.0f));
  return (uint32_t)(uint8_t)rintf(v * 255.0f);
}
__global__ __launch_bounds__(256) void tensor_to_img_u8_kernel(const float* __restrict__ t, int hw, uint8_t* __restrict__ img,
                                                               long nquads) {
  const int qpi = (hw + 3) >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (long)gridDim.x * blockDim.x) {
    const long b = i / qpi;
    const int p = (int)(i - b * qpi) * 4;
    const float* src = t + b * 3 * hw + p;
    uint8_t* px = img + (b * hw + p) * 3;
    if (p + 4 <= hw && (hw & 3) == 0) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(src);
      const f32x4 g = *reinterpret_cast<const f32x4*>(src + hw);
      const f32x4 bl = *reinterpret_cast<const f32x4*>(src + 2 * (long)hw);
      uint32_t* w = reinterpret_cast<uint32_t*>(px);
      w[0] = cf_to_u8(bl[0]) | (cf_to_u8(g[0]) << 8) | (cf_to_u8(r[0]) << 16) | (cf_to_u8(bl[1]) << 24);
      w[1] = cf_to_u8(g[1]) | (cf_to_u8(r[1]) << 8) | (cf_to_u8(bl[2]) << 16) | (cf_to_u8(g[2]) << 24);
      w[2] = cf_to_u8(r[2]) | (cf_to_u8(bl[3]) << 8) | (cf_to_u8(g[3]) << 16) | (cf_to_u8(r[3]) << 24);
    } else {
      for (int k = 0; k < 4 && p + k < hw; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) px[k * 3 + 2 - c] = (uint8_t)cf_to_u8(src[(long)c * hw + k]);
    }
  }
}

// inpainting composite (inference_inpainting.py:68-74): mask = (sum_c x == 3), i.e. pure-white pixels of the normalised
// input; out = (1-mask)*x + mask*y  ==  mask ? y : x  for finite values.
__global__ void mask_composite_kernel(const float* __restrict__ x, const float* __restrict__ y, int hw, float* __restrict__ out,
                                      long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long b = i / hw, p = i - b * hw;
  const float* xb = x + b * 3 * hw + p;
  const float x0 = xb[0], x1 = xb[hw], x2 = xb[2 * (long)hw];
  const bool m = ((x0 + x1) + x2) == 3.0f;
  const float* yb = y + b * 3 * hw + p;
  float* ob = out + b * 3 * hw + p;
  ob[0] = m ? yb[0] : x0;
  ob[hw] = m ? yb[hw] : x1;
  ob[2 * (long)hw] = m ? yb[2 * (long)hw] : x2;
}

// ---- bundled ops ----------------------------------------------------------------------------------
__global__ void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias, long numel, int c,
                                      int hw, float slope, float scale, float* __restrict__ y) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    float v = x[i];
    if (bias) v += bias[(i / hw) % c];
    y[i] = (v > 0.f ? v : v * slope) * scale;
  }
}

// ---- fused_bias_act with every mode of the reference op (fused_bias_act_kernel.cu:20-50): act 1 (linear) | 3 (leaky ReLU), grad 0 (forward) |
// 1 (first derivative, gated by the sign of `ref` = the forward OUTPUT) | 2 (second derivative: zero); element types float / IEEE half /
// bf16 (the arithmetic is done in the element type's values converted to float and rounded once on the way out, like scalar_t = half) ----
template <typename T>
__device__ __forceinline__ float fba_load(const T* p, long i) { return (float)p[i]; }
template <typename T>
__device__ __forceinline__ void fba_store(T* p, long i, float v) { p[i] = (T)v; }

template <typename T>
__global__ __launch_bounds__(256) void fused_bias_act_ex_kernel(const T* __restrict__ x, const T* __restrict__ bias, const T* __restrict__ ref,
                                                                long numel, int c, int hw, int mode, float alpha, float scale,
                                                                T* __restrict__ y) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    float v = fba_load(x, i);
    if (bias) v = (float)(T)(v + fba_load(bias, (i / hw) % c));  // (x += b in scalar_t)
    const float r = ref ? fba_load(ref, i) : 0.f;
    float o;
    switch (mode) {
      case 12:
      case 32: o = 0.f; break;
      case 30: o = v > 0.f ? v : (float)(T)(v * alpha); break;
      case 31: o = r > 0.f ? v : (float)(T)(v * alpha); break;
      default: o = v; break;  // 10, 11 and anything else: linear
    }
    fba_store(y, i, o * scale);
  }
}

// ---- upfirdn2d, LDS-tiled (upfirdn2d_kernel.cu:108-208 is the reference's tiled kernel, :251-291 its six (up, down, taps) configurations) --
// A workgroup of 256 threads owns a TH x TW tile of one output plane: the input rows / columns the tile's polyphase taps touch are staged
// in LDS with row-contiguous (coalesced) loads, the flipped FIR kernel sits in LDS zero-padded to K + UP taps per axis so the tap loops
// have compile-time trip counts and no guards, and a thread computes TW / 64 ... outputs of a row.  Planes are [nplanes][h][w] (W
// contiguous); tile sizes follow the reference's choice (16 x 64 outputs, 8 x 32 for the decimating configurations).
template <int UP, int DOWN, int K, int TH, int TW>
__global__ __launch_bounds__(256) void upfirdn2d_tiled_kernel(const float* __restrict__ x, int in_h, int in_w, const float* __restrict__ k,
                                                              int kh, int kw, int pad_x0, int pad_y0, int out_h, int out_w,
                                                              float* __restrict__ y, int tiles_x, int tiles_y) {
  constexpr int IN_H = ((TH - 1) * DOWN + K - 1) / UP + 2;
  constexpr int IN_W = ((TW - 1) * DOWN + K - 1) / UP + 2;
  constexpr int KP = K + UP;         // padded taps per axis
  constexpr int NT = (K + UP - 1) / UP;  // input samples an output touches per axis
  __shared__ float sx[IN_H][IN_W + 1];
  __shared__ float sk[KP][KP];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int plane = t / tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  for (int i = tid; i < KP * KP; i += 256) {
    const int ky = i / KP, kx = i - ky * KP;
    sk[ky][kx] = (ky < kh && kx < kw) ? k[(kh - 1 - ky) * kw + (kw - 1 - kx)] : 0.f;  // flipped FIR kernel, zero beyond its taps
  }
  // first input row / column any output of the tile can touch: ceil((o0 * DOWN - pad) / UP)
  const int by0 = oy0 * DOWN - pad_y0, bx0 = ox0 * DOWN - pad_x0;
  const int iy0 = (by0 + UP - 1 >= 0) ? (by0 + UP - 1) / UP : -((-(by0 + UP - 1) + UP - 1) / UP);
  const int ix0 = (bx0 + UP - 1 >= 0) ? (bx0 + UP - 1) / UP : -((-(bx0 + UP - 1) + UP - 1) / UP);
  const float* xp = x + (size_t)plane * in_h * in_w;
  for (int i = tid; i < IN_H * IN_W; i += 256) {
    const int ry = i / IN_W, rx = i - ry * IN_W;
    const int iy = iy0 + ry, ix = ix0 + rx;
    sx[ry][rx] = (iy >= 0 && iy < in_h && ix >= 0 && ix < in_w) ? xp[(size_t)iy * in_w + ix] : 0.f;
  }
  __syncthreads();
  for (int o = tid; o < TH * TW; o += 256) {
    const int ry = o / TW, rx = o - ry * TW;
    const int oy = oy0 + ry, ox = ox0 + rx;
    if (oy >= out_h || ox >= out_w) continue;
    const int by = oy * DOWN - pad_y0, bx = ox * DOWN - pad_x0;
    const int ky0 = ((-by) % UP + UP) % UP, kx0 = ((-bx) % UP + UP) % UP;  // first tap that lands on a real sample
    // (by + ky0) is a multiple of UP; exact division also for negatives
    const int sy = (by + ky0 >= 0 ? (by + ky0) / UP : -((-(by + ky0)) / UP)) - iy0;
    const int sxx = (bx + kx0 >= 0 ? (bx + kx0) / UP : -((-(bx + kx0)) / UP)) - ix0;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < NT; ++i) acc += sx[sy + j][sxx + i] * sk[ky0 + j * UP][kx0 + i * UP];
    y[((size_t)plane * out_h + oy) * out_w + ox] = acc;
  }
}

__global__ void upfirdn2d_kernel(const float* __restrict__ x, int in_h, int in_w, const float* __restrict__ k, int kh,
                                 int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_y0, int out_h,
                                 int out_w, float* __restrict__ y, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ox = (int)(i % out_w);
  const int oy = (int)((i / out_w) % out_h);
  const long plane = i / ((long)out_w * out_h);
  const float* xp = x + plane * in_h * in_w;
  float acc = 0.f;
  // polyphase walk: only taps that land on a real (non zero-stuffed) sample, i.e. (o*down + k - pad) % up == 0
  const int by = oy * down_y - pad_y0, bx = ox * down_x - pad_x0;
  const int ky0 = ((-by) % up_y + up_y) % up_y, kx0 = ((-bx) % up_x + up_x) % up_x;
  for (int ky = ky0; ky < kh; ky += up_y) {
    const int uy = by + ky;
    if (uy < 0) continue;
    const int iy = uy / up_y;
    if (iy >= in_h) break;
    const float* xr = xp + (size_t)iy * in_w;
    const float* kr = k + (kh - 1 - ky) * kw + (kw - 1);  // flipped FIR kernel
    for (int kx = kx0; kx < kw; kx += up_x) {
      const int ux = bx + kx;
      if (ux < 0) continue;
      const int ix = ux / up_x;
      if (ix >= in_w) break;
      acc += xr[ix] * kr[-kx];
    }
  }
  y[i] = acc;
}

// pixel-unshuffle NCHW -> zero-padded NHWC (arch_util.py:190-206); one thread per output float, channel fastest
__global__ void pixel_unshuffle_nhwc_kernel(const float* __restrict__ x, int c, int h, int w, int s, int c_pad,
                                            float* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % c_pad);
  const long pix = i / c_pad;
  const int ox = (int)(pix % w);
  const int oy = (int)((pix / w) % h);
  const long b = pix / ((long)w * h);
  float v = 0.f;
  if (k < c * s * s) {
    const int dx = k % s, dy = (k / s) % s, ci = k / (s * s);
    v = x[((b * c + ci) * (long)(h * s) + (oy * s + dy)) * (long)(w * s) + (ox * s + dx)];
  }
  out[i] = v;
}

}  // namespace

extern "C" int cf_pixel_unshuffle_nhwc(const float* x, int batch, int c, int h, int w, int s, int c_pad, float* out,
                                       cf_stream_t stream) {
  CF_REQUIRE(x && out && batch > 0 && c > 0 && h > 0 && w > 0 && s >= 1 && c_pad >= c * s * s && c_pad % 16 == 0,
             "cf_pixel_unshuffle_nhwc: bad args (c=%d s=%d c_pad=%d)", c, s, c_pad);
  const long total = (long)batch * h * w * c_pad;
  hipLaunchKernelGGL(pixel_unshuffle_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     c, h, w, s, c_pad, out, total);
  CF_CHECK_LAUNCH("cf_pixel_unshuffle_nhwc");
  return CF_OK;
}

extern "C" int cf_argmax_rows(const float* logits, int rows, int n, int64_t* idx, cf_stream_t stream) {
  CF_REQUIRE(logits && idx && rows > 0 && n > 0 && n % 4 == 0, "cf_argmax_rows: bad args (n=%d must be a multiple of 4)", n);
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, rows, n, idx);
  CF_CHECK_LAUNCH("cf_argmax_rows");
  return CF_OK;
}

extern "C" int cf_row_sqnorm(const float* x, int rows, int dim, float* out, cf_stream_t stream) {
  CF_REQUIRE(x && out && rows > 0 && dim % 4 == 0, "cf_row_sqnorm: bad args");
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows, dim, out);
  CF_CHECK_LAUNCH("cf_row_sqnorm");
  return CF_OK;
}

extern "C" int cf_vq_argmin(const float* scores, const float* zz, const float* ee, int rows, int ncodes, int64_t* idx,
                            float* dist_min, cf_stream_t stream) {
  CF_REQUIRE(scores && zz && ee && idx && rows > 0 && ncodes % 4 == 0, "cf_vq_argmin: bad args");
  hipLaunchKernelGGL(vq_argmin_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores, zz, ee, rows,
                     ncodes, idx, dist_min);
  CF_CHECK_LAUNCH("cf_vq_argmin");
  return CF_OK;
}

extern "C" int cf_codebook_gather_adain(const int64_t* idx, const float* codebook, int codebook_size, const float* lq,
                                        int batch, int ntok, int dim, int adain, float eps, float* out,
                                        cf_stream_t stream) {
  CF_REQUIRE(idx && codebook && out && (!adain || lq), "cf_codebook_gather_adain: null pointer");
  CF_REQUIRE(batch > 0 && ntok > 1 && ntok <= 8192 && dim > 0 && codebook_size > 0, "cf_codebook_gather_adain: bad dims");
  const size_t lds = (size_t)((ntok + 1) & ~1) * sizeof(int) + 4 * 8 * 32 * sizeof(double);
  hipLaunchKernelGGL(gather_adain_kernel, dim3(batch * ((dim + 31) / 32)), dim3(256), lds, (hipStream_t)stream, idx, codebook,
                     codebook_size, lq, ntok, dim, adain, eps, out);
  CF_CHECK_LAUNCH("cf_codebook_gather_adain");
  return CF_OK;
}

static int launch_transpose(const float* x, int batch, int R, int C, float* y, cf_stream_t stream, const char* name) {
  CF_REQUIRE(x && y && batch > 0 && R > 0 && C > 0 && batch < 65536, "%s: bad args", name);
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream, x, R, C,
                     y);
  CF_CHECK_LAUNCH(name);
  return CF_OK;
}
extern "C" int cf_nchw_to_nhwc(const float* x, int batch, int c, int hw, float* y, cf_stream_t stream) {
  return launch_transpose(x, batch, c, hw, y, stream, "cf_nchw_to_nhwc");
}
extern "C" int cf_nhwc_to_nchw(const float* x, int batch, int c, int hw, float* y, cf_stream_t stream) {
  return launch_transpose(x, batch, hw, c, y, stream, "cf_nhwc_to_nchw");
}

// fp32 -> bf16 copy of an activation (round to nearest even): the fp32 tensors that enter the bf16-storage part of the generator (the
// 32x32 decoder feature in front of the first bf16 Upsample, the encoder taps the fusion blocks read; cf_conv_desc.io_bf16).  Streaming:
// eight elements per thread, 32 bytes in / 16 out.
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
    const cf_u32x2 pa = cf_bf16x4_round(a), pb = cf_bf16x4_round(b);
    *reinterpret_cast<cf_u32x4*>(y + i * 8) = cf_u32x4{pa[0], pa[1], pb[0], pb[1]};
  }
}
extern "C" int cf_f32_to_bf16(const float* x, int64_t numel, void* out, cf_stream_t stream) {
  CF_REQUIRE(x && out && numel > 0 && numel % 8 == 0, "cf_f32_to_bf16: null pointer or element count %ld not a multiple of 8", (long)numel);
  const long n8 = numel / 8;
  long blocks = (n8 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<unsigned short*>(out), n8);
  CF_CHECK_LAUNCH("cf_f32_to_bf16");
  return CF_OK;
}

extern "C" int cf_img_u8_to_tensor(const uint8_t* img, int batch, int h, int w, float* out, cf_stream_t stream) {
  CF_REQUIRE(img && out && batch > 0 && h > 0 && w > 0, "cf_img_u8_to_tensor: bad args");
  const long nquads = (long)batch * ((h * w + 3) / 4);
  long blocks = (nquads + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(img_u8_to_tensor_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, h * w, out, nquads, batch);
  CF_CHECK_LAUNCH("cf_img_u8_to_tensor");
  return CF_OK;
}
extern "C" int cf_tensor_to_img_u8(const float* t, int batch, int h, int w, uint8_t* img, cf_stream_t stream) {
  CF_REQUIRE(img && t && batch > 0 && h > 0 && w > 0, "cf_tensor_to_img_u8: bad args");
  const long nquads = (long)batch * ((h * w + 3) / 4);
  long blocks = (nquads + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(tensor_to_img_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, h * w, img, nquads);
  CF_CHECK_LAUNCH("cf_tensor_to_img_u8");
  return CF_OK;
}

extern "C" int cf_mask_composite(const float* x, const float* y, int batch, int h, int w, float* out, cf_stream_t stream) {
  CF_REQUIRE(x && y && out && batch > 0 && h > 0 && w > 0, "cf_mask_composite: bad args");
  const long total = (long)batch * h * w;
  hipLaunchKernelGGL(mask_composite_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, h * w,
                     out, total);
  CF_CHECK_LAUNCH("cf_mask_composite");
  return CF_OK;
}

extern "C" int cf_fused_bias_act(const float* x, const float* bias, int64_t numel, int c, int hw, float slope, float scale,
                                 float* y, cf_stream_t stream) {
  CF_REQUIRE(x && y && numel > 0 && c > 0 && hw > 0, "cf_fused_bias_act: bad args");
  long blocks = (numel + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, (long)numel,
                     c, hw, slope, scale, y);
  CF_CHECK_LAUNCH("cf_fused_bias_act");
  return CF_OK;
}

extern "C" int cf_fused_bias_act_ex(const void* x, const void* bias, const void* ref, int64_t numel, int c, int hw, int act, int grad,
                                    float alpha, float scale, int dtype, void* y, cf_stream_t stream) {
  CF_REQUIRE(x && y && numel > 0 && c > 0 && hw > 0, "cf_fused_bias_act_ex: bad args");
  CF_REQUIRE((act == 1 || act == 3) && grad >= 0 && grad <= 2, "cf_fused_bias_act_ex: act must be 1 (linear) or 3 (leaky ReLU), grad 0..2 (got %d, %d)",
             act, grad);
  CF_REQUIRE(grad != 1 || act != 3 || ref, "cf_fused_bias_act_ex: the leaky-ReLU derivative needs ref (the forward output)");
  long blocks = (numel + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const int mode = act * 10 + grad;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case 0:
      hipLaunchKernelGGL(fused_bias_act_ex_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (const float*)bias,
                         (const float*)ref, (long)numel, c, hw, mode, alpha, scale, (float*)y);
      break;
    case 1:
      hipLaunchKernelGGL(fused_bias_act_ex_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)x,
                         (const _Float16*)bias, (const _Float16*)ref, (long)numel, c, hw, mode, alpha, scale, (_Float16*)y);
      break;
    case 2:
      hipLaunchKernelGGL(fused_bias_act_ex_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)x, (const __bf16*)bias,
                         (const __bf16*)ref, (long)numel, c, hw, mode, alpha, scale, (__bf16*)y);
      break;
    default:
      cf_set_error("cf_fused_bias_act_ex: dtype %d (0 float, 1 half, 2 bf16)", dtype);
      return CF_ERR_ARG;
  }
  CF_CHECK_LAUNCH("cf_fused_bias_act_ex");
  return CF_OK;
}

extern "C" int cf_upfirdn2d(const float* x, int nplanes, int in_h, int in_w, const float* kernel, int kh, int kw, int up_x,
                            int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float* y,
                            cf_stream_t stream) {
  CF_REQUIRE(x && y && kernel && nplanes > 0 && in_h > 0 && in_w > 0 && kh > 0 && kw > 0, "cf_upfirdn2d: bad args");
  CF_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "cf_upfirdn2d: up/down must be positive");
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  CF_REQUIRE(out_h > 0 && out_w > 0, "cf_upfirdn2d: empty output");
  // the six tiled configurations: (up, down) in {(1,1), (2,1), (1,2)} x two tap classes; everything else: the general polyphase kernel
  hipStream_t st = (hipStream_t)stream;
  if (up_x == up_y && down_x == down_y && kh <= 4 && kw <= 4) {
#define CF_UPFIRDN_TILED(UP, DOWN, K, TH, TW)                                                                                      \
  {                                                                                                                                 \
    const int tiles_x = (out_w + TW - 1) / TW, tiles_y = (out_h + TH - 1) / TH;                                                     \
    hipLaunchKernelGGL((upfirdn2d_tiled_kernel<UP, DOWN, K, TH, TW>), dim3((unsigned)((long)tiles_x * tiles_y * nplanes)), dim3(256), \
                       0, st, x, in_h, in_w, kernel, kh, kw, pad_x0, pad_y0, out_h, out_w, y, tiles_x, tiles_y);                    \
    CF_CHECK_LAUNCH("cf_upfirdn2d");                                                                                                \
    return CF_OK;                                                                                                                   \
  }
    const bool small3 = kh <= 3 && kw <= 3, small2 = kh <= 2 && kw <= 2;
    if (up_x == 1 && down_x == 1) {
      if (small3) CF_UPFIRDN_TILED(1, 1, 3, 16, 64)
      CF_UPFIRDN_TILED(1, 1, 4, 16, 64)
    }
    if (up_x == 2 && down_x == 1) {
      if (small2) CF_UPFIRDN_TILED(2, 1, 2, 16, 64)
      CF_UPFIRDN_TILED(2, 1, 4, 16, 64)
    }
    if (up_x == 1 && down_x == 2) {
      if (small2) CF_UPFIRDN_TILED(1, 2, 2, 8, 32)
      CF_UPFIRDN_TILED(1, 2, 4, 8, 32)
    }
#undef CF_UPFIRDN_TILED
  }
  const long total = (long)nplanes * out_h * out_w;
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, in_h,
                     in_w, kernel, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w, y, total);
  CF_CHECK_LAUNCH("cf_upfirdn2d");
  return CF_OK;
}
