"""CPU restatement of the paste-back step of the whole-image / video path (numpy, integer-exact where OpenCV is).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The product path (codeformer_amd/) never imports this file.

Follows facelib/utils/face_restoration_helper.py:320-362 (align_warp_face), :352-362 (get_inverse_affine) and :372-499
(paste_faces_to_input_image, square-mask branch + optional parse mask) of the reference.  The arithmetic of those lines lives in
OpenCV (cv2.warpAffine / cv2.erode / cv2.GaussianBlur / cv2.resize / cv2.invertAffineTransform), which is a dependency absent
from /root/reference AND from this container (requirements.txt: opencv-python, no pinned version), so it cannot be executed
here: **parity unpinned**.  What is restated below is OpenCV's published algorithm for each call (modules/imgproc/src/
imgwarp.cpp, resize.cpp, morph.dispatch.cpp, smooth.dispatch.cpp of the 4.x line), including its fixed-point paths:

  warpAffine, uint8, INTER_LINEAR   M is inverted in double precision; per destination pixel the source coordinate is evaluated
                                    in 10-bit fixed point (AB_BITS) from per-axis rounded terms (cvRound = round half to even),
                                    plus a rounding offset of 1/64 px, then truncated to 1/32 px (INTER_BITS = 5); the four taps are
                                    weighted with the 15-bit table (32-a)(32-b)*32 ... and the sum is rounded with (v + 2^14) >> 15;
                                    taps outside the source take borderValue
  warpAffine, float32               same coordinates; float weights (1-a/32)(1-b/32) ..., float accumulation
  erode, rectangular kernel         minimum over the k x k window anchored at k // 2; outside the image counts as +inf
  GaussianBlur(ksize, 0)            sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8, kernel exp(-x^2/(2 sigma^2)) normalised to sum 1 (float32
                                    taps), separable, BORDER_REFLECT_101
  resize, uint8, INTER_LINEAR       half-pixel centres, 11-bit weights, (v + 2^21) >> 22
  astype(np.uint8)                  truncation toward zero (what the reference's final cast does to the blended float image)
"""
import numpy as np


def invert_affine(m):
    """cv2.invertAffineTransform for a 2x3 matrix (double precision; a singular matrix gives zeros)."""
    m = np.asarray(m, dtype=np.float64).reshape(2, 3)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a12, a21 = -m[0, 1] * d, -m[1, 0] * d
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([[a11, a12, b1], [a21, a22, b2]], dtype=np.float64)


def _warp_coords(m_fwd, dw, dh):
    """Fixed-point source coordinates of every destination pixel: (ix, iy) integer parts, (a, b) 5-bit fractions."""
    inv = invert_affine(m_fwd)
    x = np.arange(dw, dtype=np.float64)
    y = np.arange(dh, dtype=np.float64)
    adelta = np.rint(inv[0, 0] * x * 1024.0).astype(np.int64)
    bdelta = np.rint(inv[1, 0] * x * 1024.0).astype(np.int64)
    x0 = np.rint((inv[0, 1] * y + inv[0, 2]) * 1024.0).astype(np.int64) + 16
    y0 = np.rint((inv[1, 1] * y + inv[1, 2]) * 1024.0).astype(np.int64) + 16
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    return X >> 5, Y >> 5, X & 31, Y & 31


def warp_affine_u8(src, m_fwd, dsize, border_value=(0, 0, 0)):
    """cv2.warpAffine(src uint8 HxWxC, M, (dw, dh), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=...)."""
    dw, dh = dsize
    h, w, c = src.shape
    ix, iy, a, b = _warp_coords(m_fwd, dw, dh)
    bv = np.asarray(border_value, dtype=np.int64)[:c]
    acc = np.zeros((dh, dw, c), dtype=np.int64)
    for dy, dx, wgt in ((0, 0, (32 - a) * (32 - b)), (0, 1, a * (32 - b)), (1, 0, (32 - a) * b), (1, 1, a * b)):
        sx, sy = ix + dx, iy + dy
        ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        tap = np.where(ok[..., None], src[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)].astype(np.int64), bv[None, None, :])
        acc += tap * (wgt * 32)[..., None]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def warp_affine_f32(src, m_fwd, dsize):
    """cv2.warpAffine of a float32 HxW image (border 0): the same 1/32-pixel coordinates, float weights."""
    dw, dh = dsize
    h, w = src.shape
    ix, iy, a, b = _warp_coords(m_fwd, dw, dh)
    fa, fb = a.astype(np.float32) / np.float32(32), b.astype(np.float32) / np.float32(32)
    out = np.zeros((dh, dw), dtype=np.float32)
    for dy, dx, wgt in ((0, 0, (1 - fa) * (1 - fb)), (0, 1, fa * (1 - fb)), (1, 0, (1 - fa) * fb), (1, 1, fa * fb)):
        sx, sy = ix + dx, iy + dy
        ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        out += np.where(ok, src[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)], np.float32(0)) * wgt.astype(np.float32)
    return out


def erode(img, k):
    """cv2.erode(img, np.ones((k, k), np.uint8)): window [-(k//2), k - 1 - k//2] on both axes, outside = +inf; k == 1: copy;
    k == 0 (an empty structuring element, reached by the reference for faces under 400 px^2): OpenCV's 3x3 default."""
    if k == 0:
        k = 3
    if k == 1:
        return img.copy()
    lo, hi = k // 2, k - 1 - k // 2
    h, w = img.shape
    pad = np.full((h + k - 1, w + k - 1), np.inf, dtype=img.dtype)
    pad[lo:lo + h, lo:lo + w] = img
    rows = np.min(np.stack([pad[:, i:i + w] for i in range(k)], 0), 0)
    return np.min(np.stack([rows[i:i + h] for i in range(k)], 0), 0)


def gaussian_kernel(ksize, sigma=0.0):
    """cv2.getGaussianKernel(ksize, sigma) as float32 taps (ksize odd).  sigma <= 0 -> 0.3*((ksize-1)*0.5 - 1) + 0.8, and for
    ksize <= 7 in that case OpenCV's fixed tables 1: [1], 3: [.25 .5 .25], 5: [.0625 .25 .375 .25 .0625],
    7: [.03125 .109375 .21875 .28125 ...]; otherwise exp(-x^2 / (2 sigma^2)) normalised to sum 1."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in small:
        return np.asarray(small[ksize], dtype=np.float32)
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i) if n > 1 else np.zeros_like(i)


def gaussian_blur(img, ksize, sigma=0.0):
    """cv2.GaussianBlur(img float32, (ksize, ksize), sigma): separable (rows then columns), BORDER_REFLECT_101, float32 accumulation."""
    k = gaussian_kernel(ksize, sigma)
    r = ksize // 2
    h, w = img.shape
    cols = _reflect101(np.arange(-r, w + r), w)
    tmp = np.zeros((h, w), dtype=np.float32)
    for i in range(ksize):
        tmp += img[:, cols[i:i + w]] * k[i]
    rows = _reflect101(np.arange(-r, h + r), h)
    out = np.zeros((h, w), dtype=np.float32)
    for i in range(ksize):
        out += tmp[rows[i:i + h]] * k[i]
    return out


def resize_linear_u8(img, dsize):
    """cv2.resize(img uint8, (dw, dh), interpolation=INTER_LINEAR): half-pixel centres, 11-bit weights, (v + 2^21) >> 22."""
    h, w = img.shape[:2]
    dw, dh = dsize

    def axis(n_src, n_dst):
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        fr = (f - i0).astype(np.float32)
        lo, hi = i0 < 0, i0 >= n_src - 1
        fr[lo | hi] = 0.0
        i0[lo] = 0
        i0[hi] = n_src - 1
        w1 = np.rint(fr.astype(np.float64) * 2048.0).astype(np.int64)
        return i0, np.minimum(i0 + 1, n_src - 1), 2048 - w1, w1

    x0, x1, wx0, wx1 = axis(w, dw)
    y0, y1, wy0, wy1 = axis(h, dh)
    s = img.astype(np.int64).reshape(h, w, -1)
    r0, r1 = s[y0], s[y1]
    top = r0[:, x0] * wx0[None, :, None] + r0[:, x1] * wx1[None, :, None]
    bot = r1[:, x0] * wx0[None, :, None] + r1[:, x1] * wx1[None, :, None]
    return ((top * wy0[:, None, None] + bot * wy1[:, None, None] + (1 << 21)) >> 22).astype(np.uint8).reshape((dh, dw) + img.shape[2:])


def align_warp_face(frame, affine, face_size=(512, 512)):
    """face_restoration_helper.py:343-344: cv2.warpAffine(input_img, affine_matrix, face_size, borderMode=BORDER_CONSTANT,
    borderValue=(135, 133, 132))."""
    return warp_affine_u8(frame, affine, face_size, border_value=(135, 133, 132))


def paste_faces(frame, restored_faces, affines, upscale=1, face_size=(512, 512), parse_masks=None, return_float=False):
    """paste_faces_to_input_image (face_restoration_helper.py:372-499), upsample_img=None, draw_box=False, face_upsampler=None.
    frame: uint8 HxWx3 BGR; restored_faces: uint8 512x512x3 each; affines: the 2x3 alignment matrices (frame -> face).
    parse_masks: optional per-face float32 512x512 soft masks in [0, 1] ALREADY blurred / border-cleared (:466-477); they are
    resized to face_size, warped with flags=3 (treated as INTER_LINEAR here, see DESIGN) and fused as :479-483."""
    h, w = frame.shape[:2]
    h_up, w_up = int(h * upscale), int(w * upscale)
    img = resize_linear_u8(frame, (w_up, h_up)).astype(np.float32) if (w_up, h_up) != (w, h) else frame.astype(np.float32)
    first = True
    for k, (face, aff) in enumerate(zip(restored_faces, affines)):
        inv = invert_affine(aff) * upscale                       # get_inverse_affine (:352-356)
        inv[:, 2] += 0.5 * upscale if upscale > 1 else 0         # :393-398
        inv_restored = warp_affine_u8(face, _as_forward(inv), (w_up, h_up))
        inv_mask = warp_affine_f32(np.ones(face_size[::-1], np.float32), _as_forward(inv), (w_up, h_up))
        inv_mask_erosion = erode(inv_mask, int(2 * upscale))
        pasted = inv_mask_erosion[:, :, None] * inv_restored.astype(np.float32)
        total_face_area = np.sum(inv_mask_erosion)
        w_edge = int(total_face_area ** 0.5) // 20
        inv_mask_center = erode(inv_mask_erosion, w_edge * 2)
        inv_soft_mask = gaussian_blur(inv_mask_center, w_edge * 2 + 1)
        if parse_masks is not None and parse_masks[k] is not None:
            pm = warp_affine_f32(parse_masks[k].astype(np.float32), _as_forward(inv), (w_up, h_up))
            inv_soft_mask = np.where(pm < inv_soft_mask, pm, inv_soft_mask)
        m = inv_soft_mask[:, :, None]
        if first and img.dtype != np.float32:
            img = img.astype(np.float32)
        img = m * pasted + (1 - m) * img
        first = False
    return img if return_float else img.astype(np.uint8)


MASK_COLORMAP = (0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 0, 255, 0, 0, 0)


def parse_soft_mask(labels):
    """face_restoration_helper.py:466-481 for one (512,512) label map: colour map, GaussianBlur((101,101), 11) twice, 10-pixel borders
    cleared, / 255.  (float32 here; the reference's array is float64 -- see the module header of codeformer_amd/facelib/paste.py.)"""
    m = np.zeros(labels.shape, dtype=np.float32)
    for idx, color in enumerate(MASK_COLORMAP):
        m[labels == idx] = color
    m = gaussian_blur(gaussian_blur(m, 101, 11.0), 101, 11.0)
    m[:10, :] = 0
    m[-10:, :] = 0
    m[:, :10] = 0
    m[:, -10:] = 0
    return m * np.float32(1.0 / 255.0)


def _as_forward(inv):
    """The reference hands the INVERSE affine (face -> frame) to cv2.warpAffine as its forward matrix M."""
    return np.asarray(inv, dtype=np.float64)


def similarity_from_points(src, dst):
    """Least-squares similarity transform (rotation + uniform scale + translation) mapping src -> dst points (Nx2): the model
    cv2.estimateAffinePartial2D fits (face_restoration_helper.py:329; its LMEDS consensus step equals the least-squares fit
    when every point is an inlier).  Used by tests / synthetic benches to make plausible alignment matrices."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    ms, md = src.mean(0), dst.mean(0)
    s, d = src - ms, dst - md
    a = (s * d).sum() / (s * s).sum()
    b = (s[:, 0] * d[:, 1] - s[:, 1] * d[:, 0]).sum() / (s * s).sum()
    r = np.array([[a, -b], [b, a]])
    t = md - r @ ms
    return np.array([[a, -b, t[0]], [b, a, t[1]]], dtype=np.float64)
