// Paste-back of restored faces into the (upsampled) frame, and the alignment warp that cuts the crops -- the stage around
// CodeFormer.forward on the whole-image / video path (reference: facelib/utils/face_restoration_helper.py:320-362 align_warp_face,
// :372-499 paste_faces_to_input_image).  The reference runs these as OpenCV calls on the host over the WHOLE upsampled frame per
// face; here each is an HBM-bound kernel over the face's bounding box only, on tensors that stay on the device:
//
//   cv2.warpAffine (uint8 / float32, INTER_LINEAR, constant border)  -> warp_u8_kernel / warp_f32_kernel
//   cv2.erode (rectangular kernel)                                    -> minfilter_kernel (separable: rows, then columns)
//   cv2.GaussianBlur(ksize, 0)                                        -> gauss_kernel (separable, BORDER_REFLECT_101 of the frame)
//   np.sum (face area)                                                -> sum_kernel (fp64 partials, fixed order)
//   inv_soft_mask*pasted + (1-inv_soft_mask)*img, parse-mask fusion   -> blend_kernel (samples the restored face itself)
//   cv2.resize(INTER_LINEAR) of the background, final astype(uint8)   -> resize_u8_kernel, trunc_u8_kernel
//   cv2.resize(INTER_AREA) of the frame for the detector (:208-215)    -> resize_area_u8_kernel
//
// The arithmetic follows OpenCV's fixed-point definitions (restated with citations in oracle/paste_oracle.py): source coordinates
// in 10-bit fixed point from per-axis rounded terms, +1/64 px, truncated to 1/32 px; uint8 taps weighted with the 15-bit table and
// rounded with (v + 2^14) >> 15; float taps with float weights in the order 00, 01, 10, 11.  All kernels are one thread per output
// pixel (or 4-byte group), coalesced along the row; every buffer is at most a few MB, so they are latency- rather than
// bandwidth-limited and are batched per frame by the host (codeformer_amd/facelib/paste.py).
#include "cf_common.h"

// numpy / OpenCV round every multiply and add separately.  HIP's __fmul_rn / __fadd_rn are plain operators defined in a header
// (compiled under hipcc's default -ffp-contract=fast, so their results still fuse into FMAs); the helpers below are defined
// lexically under the pragma and carry no contraction permission.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

struct Inv6 {
  double m[6];  // dst (x, y) -> src: sx = m0*x + m1*y + m2 ; sy = m3*x + m4*y + m5  (already inverted on the host, in double)
};

// OpenCV's WarpAffineInvoker coordinates: X = (cvRound((m1*y + m2)*1024) + 16 + cvRound(m0*x*1024)) >> 5, fraction = X & 31
__device__ __forceinline__ void warp_coord(const Inv6& v, int x, int y, int& ix, int& iy, int& a, int& b) {
  const long long X = (__double2ll_rn((v.m[1] * (double)y + v.m[2]) * 1024.0) + 16 + __double2ll_rn(v.m[0] * (double)x * 1024.0)) >> 5;
  const long long Y = (__double2ll_rn((v.m[4] * (double)y + v.m[5]) * 1024.0) + 16 + __double2ll_rn(v.m[3] * (double)x * 1024.0)) >> 5;
  ix = (int)(X >> 5);
  iy = (int)(Y >> 5);
  a = (int)(X & 31);
  b = (int)(Y & 31);
}

// dst canvas [n][dh][dw][3] u8, region (rx, ry, rw, rh) of each; src [sh][sw][3] u8 (src_stride = 0: one source for every item)
__global__ __launch_bounds__(256) void warp_u8_kernel(const uint8_t* __restrict__ src, long src_stride, int sh, int sw,
                                                      const Inv6* __restrict__ inv, uint8_t* __restrict__ dst, int dh, int dw, int rx,
                                                      int ry, int rw, int rh, int bv0, int bv1, int bv2) {
  const int n = blockIdx.y;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rw * rh) return;
  const int x = rx + (int)(i % rw), y = ry + (int)(i / rw);
  const Inv6 v = inv[n];
  int ix, iy, a, b;
  warp_coord(v, x, y, ix, iy, a, b);
  const uint8_t* s = src + n * src_stride;
  const int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32, w11 = a * b * 32;
  const bool x0 = ix >= 0 && ix < sw, x1 = ix + 1 >= 0 && ix + 1 < sw, y0 = iy >= 0 && iy < sh, y1 = iy + 1 >= 0 && iy + 1 < sh;
  const int bv[3] = {bv0, bv1, bv2};
  uint8_t* o = dst + (((long)n * dh + y) * dw + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int p00 = (x0 && y0) ? s[((long)iy * sw + ix) * 3 + c] : bv[c];
    const int p01 = (x1 && y0) ? s[((long)iy * sw + ix + 1) * 3 + c] : bv[c];
    const int p10 = (x0 && y1) ? s[((long)(iy + 1) * sw + ix) * 3 + c] : bv[c];
    const int p11 = (x1 && y1) ? s[((long)(iy + 1) * sw + ix + 1) * 3 + c] : bv[c];
    o[c] = (uint8_t)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
  }
}

// float32 single-channel source [sh][sw] -> compact region buffer [rh][rw] (border 0)
__global__ __launch_bounds__(256) void warp_f32_kernel(const float* __restrict__ src, int sh, int sw, Inv6 v, float* __restrict__ dst,
                                                       int rx, int ry, int rw, int rh) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rw * rh) return;
  const int x = rx + (int)(i % rw), y = ry + (int)(i / rw);
  int ix, iy, a, b;
  warp_coord(v, x, y, ix, iy, a, b);
  const float fa = (float)a / 32.0f, fb = (float)b / 32.0f;
  const bool x0 = ix >= 0 && ix < sw, x1 = ix + 1 >= 0 && ix + 1 < sw, y0 = iy >= 0 && iy < sh, y1 = iy + 1 >= 0 && iy + 1 < sh;
  float acc = 0.f;
  acc = add_rn(acc, mul_rn((x0 && y0) ? src[(long)iy * sw + ix] : 0.f, mul_rn(1.f - fa, 1.f - fb)));
  acc = add_rn(acc, mul_rn((x1 && y0) ? src[(long)iy * sw + ix + 1] : 0.f, mul_rn(fa, 1.f - fb)));
  acc = add_rn(acc, mul_rn((x0 && y1) ? src[(long)(iy + 1) * sw + ix] : 0.f, mul_rn(1.f - fa, fb)));
  acc = add_rn(acc, mul_rn((x1 && y1) ? src[(long)(iy + 1) * sw + ix + 1] : 0.f, mul_rn(fa, fb)));
  dst[i] = acc;
}

// 1-D minimum over [-lo, +hi] along x (axis 0) or y (axis 1); samples outside the region do not take part (+inf)
__global__ __launch_bounds__(256) void minfilter_kernel(const float* __restrict__ in, float* __restrict__ out, int rh, int rw, int lo, int hi,
                                                        int axis) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rw * rh) return;
  const int x = (int)(i % rw), y = (int)(i / rw);
  float m = INFINITY;
  if (axis == 0) {
    const int b = max(x - lo, 0), e = min(x + hi, rw - 1);
    for (int k = b; k <= e; ++k) m = fminf(m, in[(long)y * rw + k]);
  } else {
    const int b = max(y - lo, 0), e = min(y + hi, rh - 1);
    for (int k = b; k <= e; ++k) m = fminf(m, in[(long)k * rw + x]);
  }
  out[i] = m;
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n <= 1) return 0;
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// 1-D Gaussian along x (axis 0) or y (axis 1).  The region is a window (rx, ry) of a cw x ch frame: indices are reflected (101) at
// the FRAME border like OpenCV does; frame positions outside the region hold an exact 0 (the host sizes the region so).
__global__ __launch_bounds__(256) void gauss_kernel(const float* __restrict__ in, float* __restrict__ out, int rh, int rw, int rx, int ry,
                                                    int ch, int cw, const float* __restrict__ taps, int ksize, int axis) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rw * rh) return;
  const int x = (int)(i % rw), y = (int)(i / rw);
  const int r = ksize / 2;
  float acc = 0.f;
  for (int k = 0; k < ksize; ++k) {
    float v = 0.f;
    if (axis == 0) {
      const int fx = reflect101(rx + x + k - r, cw) - rx;
      if (fx >= 0 && fx < rw) v = in[(long)y * rw + fx];
    } else {
      const int fy = reflect101(ry + y + k - r, ch) - ry;
      if (fy >= 0 && fy < rh) v = in[(long)fy * rw + x];
    }
    acc = add_rn(acc, mul_rn(v, taps[k]));
  }
  out[i] = acc;
}

// fp64 partial sums in a fixed order: block b adds elements b*chunk .. (b+1)*chunk-1 (tree inside the block); the host adds the partials
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, long n, long chunk, double* __restrict__ partial) {
  __shared__ double red[256];
  const long lo = (long)blockIdx.x * chunk, hi = min(n, lo + chunk);
  double s = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) s += (double)x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// canvas[y][x][c] = m * (ero * face(x, y)[c]) + (1 - m) * canvas  over the region; m = soft mask, optionally fused with the parse mask
// (m = parse < soft ? parse : soft, face_restoration_helper.py:481-483); separately rounded float32 operations as numpy evaluates them
__global__ __launch_bounds__(256) void blend_kernel(float* __restrict__ canvas, int ch, int cw, const uint8_t* __restrict__ face, int fh,
                                                    int fw, Inv6 v, const float* __restrict__ ero, const float* __restrict__ soft,
                                                    const float* __restrict__ parse, int rx, int ry, int rw, int rh) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rw * rh) return;
  const int x = rx + (int)(i % rw), y = ry + (int)(i / rw);
  float m = soft[i];
  if (parse) {
    const float p = parse[i];
    m = p < m ? p : m;
  }
  int ix, iy, a, b;
  warp_coord(v, x, y, ix, iy, a, b);
  const int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32, w11 = a * b * 32;
  const bool x0 = ix >= 0 && ix < fw, x1 = ix + 1 >= 0 && ix + 1 < fw, y0 = iy >= 0 && iy < fh, y1 = iy + 1 >= 0 && iy + 1 < fh;
  const float e = ero[i];
  float* o = canvas + ((long)y * cw + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int p00 = (x0 && y0) ? face[((long)iy * fw + ix) * 3 + c] : 0;
    const int p01 = (x1 && y0) ? face[((long)iy * fw + ix + 1) * 3 + c] : 0;
    const int p10 = (x0 && y1) ? face[((long)(iy + 1) * fw + ix) * 3 + c] : 0;
    const int p11 = (x1 && y1) ? face[((long)(iy + 1) * fw + ix + 1) * 3 + c] : 0;
    const float restored = (float)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
    const float pasted = mul_rn(e, restored);
    o[c] = add_rn(mul_rn(m, pasted), mul_rn(sub_rn(1.0f, m), o[c]));
  }
}

// cv2.resize(uint8, INTER_LINEAR) -> float32 canvas (exact integers): 11-bit weights per axis, (v + 2^21) >> 22
__device__ __forceinline__ void resize_axis(int d, int n_src, int n_dst, int& i0, int& i1, int& w1) {
  const double f = ((double)d + 0.5) * ((double)n_src / (double)n_dst) - 0.5;
  int i = (int)floor(f);
  float fr = (float)(f - (double)i);
  if (i < 0) {
    i = 0;
    fr = 0.f;
  }
  if (i >= n_src - 1) {
    i = n_src - 1;
    fr = 0.f;
  }
  i0 = i;
  i1 = min(i + 1, n_src - 1);
  w1 = (int)__double2ll_rn((double)fr * 2048.0);
}
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, float* __restrict__ dst, int dh,
                                                        int dw) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)dw * dh) return;
  const int x = (int)(i % dw), y = (int)(i / dw);
  int x0, x1, wx1, y0, y1, wy1;
  resize_axis(x, sw, dw, x0, x1, wx1);
  resize_axis(y, sh, dh, y0, y1, wy1);
  const long long wx0 = 2048 - wx1, wy0 = 2048 - wy1;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const long long top = src[((long)y0 * sw + x0) * 3 + c] * wx0 + src[((long)y0 * sw + x1) * 3 + c] * (long long)wx1;
    const long long bot = src[((long)y1 * sw + x0) * 3 + c] * wx0 + src[((long)y1 * sw + x1) * 3 + c] * (long long)wx1;
    dst[i * 3 + c] = (float)((top * wy0 + bot * wy1 + (1 << 21)) >> 22);
  }
}

// cv2.resize(INTER_AREA), uint8, shrinking on both axes (FaceRestoreHelper.get_face_landmarks_5 reduces frames to the detector's
// working size with it, face_restoration_helper.py:208-215).  OpenCV's rule (computeResizeAreaTab / ResizeArea_Invoker), evaluated per
// output pixel: the covered source interval [d*s, (d+1)*s) as [fractional first cell] + whole cells + [fractional last cell], weights
// in double rounded to float; float32 accumulation in source order -- row sums first (buf += src * alpha), then sum += beta * buf -- with
// separately rounded multiplies and adds; cvRound at the end.  Integer ratios: integer box sums, (sum + 2) >> 2 for 2 x 2, else
// cvRound(sum * (1.f / area)).  Restated in numpy as codeformer_amd.utils.img_util.resize_area (the test oracle of this kernel).
struct AreaTaps {
  int first;        // first source index touched
  int n;            // number of taps
  float a0, am, a1; // weight of the fractional first cell (0: none), of a whole cell, of the fractional last cell (0: none)
  bool has0, has1;
};
__device__ __forceinline__ AreaTaps area_taps(int d, double scale, int nsrc) {
  const double f1 = (double)d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, (double)nsrc - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = s2 < nsrc - 1 ? s2 : nsrc - 1;
  s1 = s1 < s2 ? s1 : s2;
  AreaTaps t;
  t.has0 = (double)s1 - f1 > 1e-3;
  t.has1 = f2 - (double)s2 > 1e-3;
  t.a0 = t.has0 ? (float)(((double)s1 - f1) / cell) : 0.f;
  t.am = (float)(1.0 / cell);
  t.a1 = t.has1 ? (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell) : 0.f;
  t.first = t.has0 ? s1 - 1 : s1;
  t.n = (s2 - s1) + (t.has0 ? 1 : 0) + (t.has1 ? 1 : 0);
  return t;
}
__device__ __forceinline__ float area_weight(const AreaTaps& t, int k) {
  if (k == 0 && t.has0) return t.a0;
  if (k == t.n - 1 && t.has1) return t.a1;
  return t.am;
}
__global__ __launch_bounds__(256) void resize_area_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh,
                                                             int dw, double scale_x, double scale_y, int kx, int ky) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)dw * dh) return;
  const int x = (int)(i % dw), y = (int)(i / dw);
  if (kx > 0) {  // integer ratios: box sums
    int sum[3] = {0, 0, 0};
    for (int r = 0; r < ky; ++r)
      for (int q = 0; q < kx; ++q) {
        const uint8_t* p = src + ((long)(y * ky + r) * sw + (x * kx + q)) * 3;
        sum[0] += p[0];
        sum[1] += p[1];
        sum[2] += p[2];
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int v;
      if (kx == 2 && ky == 2) v = (sum[c] + 2) >> 2;
      else v = (int)__float2ll_rn(mul_rn((float)sum[c], 1.f / (float)(kx * ky)));
      dst[i * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    return;
  }
  const AreaTaps tx = area_taps(x, scale_x, sw), ty = area_taps(y, scale_y, sh);
  float sum[3] = {0.f, 0.f, 0.f};
  for (int r = 0; r < ty.n; ++r) {
    const float beta = area_weight(ty, r);
    float buf[3] = {0.f, 0.f, 0.f};
    for (int q = 0; q < tx.n; ++q) {
      const float alpha = area_weight(tx, q);
      const uint8_t* p = src + ((long)(ty.first + r) * sw + (tx.first + q)) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) buf[c] = add_rn(buf[c], mul_rn((float)p[c], alpha));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) sum[c] = add_rn(sum[c], mul_rn(buf[c], beta));
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const long long v = __float2ll_rn(sum[c]);
    dst[i * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

// ndarray.astype(np.uint8) of a non-negative float image: truncation toward zero (values are in [0, 255] by construction)
__global__ __launch_bounds__(256) void trunc_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (uint8_t)(int)src[i];
}

// parse-map colouring (face_restoration_helper.py:468-471): out = lut[label]
struct Lut32 {
  float v[32];
};
__global__ __launch_bounds__(256) void label_lut_kernel(const int64_t* __restrict__ labels, long n, Lut32 lut, int nlut, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t l = labels[i];
  out[i] = (l >= 0 && l < nlut) ? lut.v[l] : 0.f;
}
// x[b][y][x] = inside the border frame ? x * scale : 0   (face_restoration_helper.py:476-481: clear 10-pixel borders, / 255)
__global__ __launch_bounds__(256) void scale_clear_border_kernel(float* __restrict__ x, int h, int w, long n, int border, float scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int px = (int)(i % w), py = (int)((i / w) % h);
  const bool inside = px >= border && px < w - border && py >= border && py < h - border;
  x[i] = inside ? mul_rn(x[i], scale) : 0.f;
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }
inline Inv6 load6(const double* m) {
  Inv6 v;
  for (int i = 0; i < 6; ++i) v.m[i] = m[i];
  return v;
}

}  // namespace

extern "C" int cf_warp_affine_u8(const uint8_t* src, int64_t src_stride, int sh, int sw, const double* inv_dev, int n, uint8_t* dst,
                                 int dh, int dw, int rx, int ry, int rw, int rh, int b0, int b1, int b2, cf_stream_t stream) {
  CF_REQUIRE(src && inv_dev && dst && n > 0 && sh > 0 && sw > 0, "cf_warp_affine_u8: bad args");
  CF_REQUIRE(rx >= 0 && ry >= 0 && rw > 0 && rh > 0 && rx + rw <= dw && ry + rh <= dh, "cf_warp_affine_u8: region %d,%d %dx%d outside %dx%d",
             rx, ry, rw, rh, dw, dh);
  hipLaunchKernelGGL(warp_u8_kernel, dim3(nblk((long)rw * rh), n), dim3(256), 0, (hipStream_t)stream, src, (long)src_stride, sh, sw,
                     reinterpret_cast<const Inv6*>(inv_dev), dst, dh, dw, rx, ry, rw, rh, b0, b1, b2);
  CF_CHECK_LAUNCH("cf_warp_affine_u8");
  return CF_OK;
}

extern "C" int cf_warp_affine_f32(const float* src, int sh, int sw, const double* inv_host, float* dst_region, int rx, int ry, int rw,
                                  int rh, cf_stream_t stream) {
  CF_REQUIRE(src && inv_host && dst_region && sh > 0 && sw > 0 && rw > 0 && rh > 0, "cf_warp_affine_f32: bad args");
  hipLaunchKernelGGL(warp_f32_kernel, dim3(nblk((long)rw * rh)), dim3(256), 0, (hipStream_t)stream, src, sh, sw, load6(inv_host),
                     dst_region, rx, ry, rw, rh);
  CF_CHECK_LAUNCH("cf_warp_affine_f32");
  return CF_OK;
}

extern "C" int cf_erode_f32(const float* in, float* tmp, float* out, int rh, int rw, int k, cf_stream_t stream) {
  CF_REQUIRE(in && tmp && out && rh > 0 && rw > 0 && k >= 0, "cf_erode_f32: bad args");
  if (k == 0) k = 3;  // an empty structuring element makes cv2.erode use its 3x3 default
  const int lo = k / 2, hi = k - 1 - k / 2;
  hipLaunchKernelGGL(minfilter_kernel, dim3(nblk((long)rw * rh)), dim3(256), 0, (hipStream_t)stream, in, tmp, rh, rw, lo, hi, 0);
  hipLaunchKernelGGL(minfilter_kernel, dim3(nblk((long)rw * rh)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, out, rh, rw, lo,
                     hi, 1);
  CF_CHECK_LAUNCH("cf_erode_f32");
  return CF_OK;
}

extern "C" int cf_gaussian_blur_f32(const float* in, float* tmp, float* out, int rh, int rw, int rx, int ry, int ch, int cw,
                                    const float* taps_dev, int ksize, cf_stream_t stream) {
  CF_REQUIRE(in && tmp && out && taps_dev && rh > 0 && rw > 0 && ksize > 0 && (ksize & 1), "cf_gaussian_blur_f32: bad args (ksize %d)", ksize);
  CF_REQUIRE(rx >= 0 && ry >= 0 && rx + rw <= cw && ry + rh <= ch, "cf_gaussian_blur_f32: region outside the frame");
  hipLaunchKernelGGL(gauss_kernel, dim3(nblk((long)rw * rh)), dim3(256), 0, (hipStream_t)stream, in, tmp, rh, rw, rx, ry, ch, cw, taps_dev,
                     ksize, 0);
  hipLaunchKernelGGL(gauss_kernel, dim3(nblk((long)rw * rh)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, out, rh, rw, rx, ry, ch,
                     cw, taps_dev, ksize, 1);
  CF_CHECK_LAUNCH("cf_gaussian_blur_f32");
  return CF_OK;
}

extern "C" int cf_sum_f32(const float* x, int64_t n, double* partials64, cf_stream_t stream) {
  CF_REQUIRE(x && partials64 && n > 0, "cf_sum_f32: bad args");
  const long chunk = (n + 63) / 64;
  hipLaunchKernelGGL(sum_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, x, (long)n, chunk, partials64);
  CF_CHECK_LAUNCH("cf_sum_f32");
  return CF_OK;
}

extern "C" int cf_paste_blend(float* canvas, int ch, int cw, const uint8_t* face, int fh, int fw, const double* inv_host,
                              const float* ero_region, const float* soft_region, const float* parse_region, int rx, int ry, int rw, int rh,
                              cf_stream_t stream) {
  CF_REQUIRE(canvas && face && inv_host && ero_region && soft_region && fh > 0 && fw > 0, "cf_paste_blend: bad args");
  CF_REQUIRE(rx >= 0 && ry >= 0 && rw > 0 && rh > 0 && rx + rw <= cw && ry + rh <= ch, "cf_paste_blend: region outside the frame");
  hipLaunchKernelGGL(blend_kernel, dim3(nblk((long)rw * rh)), dim3(256), 0, (hipStream_t)stream, canvas, ch, cw, face, fh, fw,
                     load6(inv_host), ero_region, soft_region, parse_region, rx, ry, rw, rh);
  CF_CHECK_LAUNCH("cf_paste_blend");
  return CF_OK;
}

extern "C" int cf_resize_linear_u8(const uint8_t* src, int sh, int sw, float* dst, int dh, int dw, cf_stream_t stream) {
  CF_REQUIRE(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0, "cf_resize_linear_u8: bad args");
  if (sh == dh && sw == dw)
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(nblk((long)dh * dw * 3)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)dh * dw * 3);
  else
    hipLaunchKernelGGL(resize_u8_kernel, dim3(nblk((long)dh * dw)), dim3(256), 0, (hipStream_t)stream, src, sh, sw, dst, dh, dw);
  CF_CHECK_LAUNCH("cf_resize_linear_u8");
  return CF_OK;
}

extern "C" int cf_resize_area_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, cf_stream_t stream) {
  CF_REQUIRE(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0, "cf_resize_area_u8: bad args");
  CF_REQUIRE(dh <= sh && dw <= sw, "cf_resize_area_u8: INTER_AREA is built for shrinking (%dx%d -> %dx%d)", sw, sh, dw, dh);
  const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  const bool integer = sw % dw == 0 && sh % dh == 0;
  hipLaunchKernelGGL(resize_area_u8_kernel, dim3(nblk((long)dh * dw)), dim3(256), 0, (hipStream_t)stream, src, sh, sw, dst, dh, dw, scale_x,
                     scale_y, integer ? sw / dw : 0, integer ? sh / dh : 0);
  CF_CHECK_LAUNCH("cf_resize_area_u8");
  return CF_OK;
}

extern "C" int cf_f32_to_u8_trunc(const float* src, int64_t n, uint8_t* dst, cf_stream_t stream) {
  CF_REQUIRE(src && dst && n > 0, "cf_f32_to_u8_trunc: bad args");
  hipLaunchKernelGGL(trunc_u8_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
  CF_CHECK_LAUNCH("cf_f32_to_u8_trunc");
  return CF_OK;
}

extern "C" int cf_label_lut_f32(const int64_t* labels, int64_t n, const float* lut_host, int nlut, float* out, cf_stream_t stream) {
  CF_REQUIRE(labels && lut_host && out && n > 0 && nlut > 0 && nlut <= 32, "cf_label_lut_f32: bad args (nlut %d)", nlut);
  Lut32 lut;
  for (int i = 0; i < 32; ++i) lut.v[i] = i < nlut ? lut_host[i] : 0.f;
  hipLaunchKernelGGL(label_lut_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, labels, (long)n, lut, nlut, out);
  CF_CHECK_LAUNCH("cf_label_lut_f32");
  return CF_OK;
}

extern "C" int cf_scale_clear_border_f32(float* x, int batch, int h, int w, int border, float scale, cf_stream_t stream) {
  CF_REQUIRE(x && batch > 0 && h > 0 && w > 0 && border >= 0, "cf_scale_clear_border_f32: bad args");
  const long n = (long)batch * h * w;
  hipLaunchKernelGGL(scale_clear_border_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, h, w, n, border, scale);
  CF_CHECK_LAUNCH("cf_scale_clear_border_f32");
  return CF_OK;
}
