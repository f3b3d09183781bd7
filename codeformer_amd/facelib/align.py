"""cv2.estimateAffinePartial2D(src, dst, method=cv2.LMEDS) -- the alignment fit of FaceRestoreHelper.align_warp_face
(facelib/utils/face_restoration_helper.py:329) -- restated for a host without OpenCV.

Model: 4-dof similarity  [[a, -b, tx], [b, a, ty]]  mapping landmarks onto the face template.
Algorithm as OpenCV 4.x runs it with the function's default arguments (maxIters 2000, confidence 0.99, refineIters 10):
  1. points are converted to float32;
  2. least-median-of-squares over minimal samples: niters = round(log(1 - 0.99) / log(1 - (1 - 0.45)^2)) = 13 two-point subsets drawn
     with OpenCV's multiply-with-carry generator seeded with 2^64 - 1 (`uniform(0, n)` = next() % n, duplicates redrawn); each subset
     gives the exact two-point similarity; its score is the median (element n/2 of the sorted list) of the squared residuals, evaluated
     in float32 with the model rounded to float32; the subset with the smallest median wins;
  3. inliers: squared residual <= (2.5 * 1.4826 * (1 + 5 / (n - 2)) * sqrt(median))^2 (at least 0.001^2);
  4. when more than two inliers remain, the model is refined on the inliers by Levenberg-Marquardt on the reprojection error.  The
     residual is linear in (a, b, tx, ty), so the LM iterations converge to the linear least-squares similarity of the inliers;
     here that solution is computed in closed form (float64).
PARITY UNPINNED: OpenCV is absent from the reference tree and from the build container; the restatement follows OpenCV's published
source as cited above and is checked for self-consistency only (tests/test_detection.py).  With outlier-free landmarks every point is
an inlier and the result is the least-squares similarity of all five points, independent of the sampling.
"""
import math

import numpy as np

_COEFF = 4164903690
_MASK32 = 0xFFFFFFFF


class _Rng:
    """cv::RNG: state <- (uint32) state * 4164903690 + (state >> 32); next() = low 32 bits."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state

    def uniform(self, lo, hi):
        self.state = ((self.state & _MASK32) * _COEFF + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return lo if lo == hi else lo + (self.state & _MASK32) % (hi - lo)


def _two_point_similarity(p, q):
    """Exact similarity through p[0]->q[0], p[1]->q[1] (float64), as a 2x3 matrix."""
    (x1, y1), (x2, y2) = p
    (X1, Y1), (X2, Y2) = q
    dx, dy, dX, dY = x1 - x2, y1 - y2, X1 - X2, Y1 - Y2
    d = 1.0 / (dx * dx + dy * dy)
    a = d * (dX * dx + dY * dy)
    b = d * (dY * dx - dX * dy)
    cr = x1 * y2 - x2 * y1
    tx = d * (dY * cr - (X1 * y2 - X2 * y1) * dy - (X1 * x2 - X2 * x1) * dx)
    ty = d * (-dX * cr - (Y1 * x2 - Y2 * x1) * dx - (Y1 * y2 - Y2 * y1) * dy)
    return np.array([[a, -b, tx], [b, a, ty]], dtype=np.float64)


def _residuals_f32(model, src, dst):
    m = model.astype(np.float32)
    ex = m[0, 0] * src[:, 0] + m[0, 1] * src[:, 1] + m[0, 2] - dst[:, 0]
    ey = m[1, 0] * src[:, 0] + m[1, 1] * src[:, 1] + m[1, 2] - dst[:, 1]
    return (ex * ex + ey * ey).astype(np.float32)


def least_squares_similarity(src, dst):
    """argmin over (a, b, tx, ty) of sum |[[a,-b],[b,a]] p + t - q|^2 (float64, closed form)."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    mp, mq = src.mean(axis=0), dst.mean(axis=0)
    p, q = src - mp, dst - mq
    den = float((p * p).sum())
    a = float((p * q).sum()) / den
    b = float((p[:, 0] * q[:, 1] - p[:, 1] * q[:, 0]).sum()) / den
    tx = mq[0] - (a * mp[0] - b * mp[1])
    ty = mq[1] - (b * mp[0] + a * mp[1])
    return np.array([[a, -b, tx], [b, a, ty]], dtype=np.float64)


def estimate_affine_partial_2d(src, dst, confidence=0.99, max_iters=2000, refine=True):
    """Returns (2x3 float64 matrix or None, inlier mask uint8 (n,1)) like cv2.estimateAffinePartial2D(src, dst, method=cv2.LMEDS)."""
    src = np.asarray(src, dtype=np.float32).reshape(-1, 2)
    dst = np.asarray(dst, dtype=np.float32).reshape(-1, 2)
    n = src.shape[0]
    if n != dst.shape[0] or n < 2:
        return None, np.zeros((n, 1), dtype=np.uint8)
    s64, d64 = src.astype(np.float64), dst.astype(np.float64)
    if n == 2:
        return _two_point_similarity(s64, d64), np.ones((n, 1), dtype=np.uint8)
    num, den = math.log(max(1.0 - confidence, 2.2250738585072014e-308)), math.log(1.0 - (1.0 - 0.45) ** 2)
    niters = max_iters if -num >= max_iters * -den else int(round(num / den))
    niters = max(niters, 3)
    rng, best, best_median = _Rng(), None, math.inf
    for _ in range(niters):
        idx = []
        while len(idx) < 2:
            k = rng.uniform(0, n)
            if k not in idx:
                idx.append(k)
        if s64[idx[0], 0] == s64[idx[1], 0] and s64[idx[0], 1] == s64[idx[1], 1]:
            continue
        model = _two_point_similarity(s64[idx], d64[idx])
        median = float(np.sort(_residuals_f32(model, src, dst))[n // 2])
        if median < best_median:
            best_median, best = median, model
    if best is None:
        return None, np.zeros((n, 1), dtype=np.uint8)
    sigma = max(2.5 * 1.4826 * (1.0 + 5.0 / (n - 2)) * math.sqrt(best_median), 0.001)
    inl = _residuals_f32(best, src, dst) <= np.float32(sigma * sigma)
    if refine and int(inl.sum()) > 2:
        best = least_squares_similarity(s64[inl], d64[inl])
    return best, inl.astype(np.uint8).reshape(-1, 1)
