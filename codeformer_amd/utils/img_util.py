"""Image <-> tensor helpers of the tensor boundary (API of the reference's basicsr/utils/img_util.py:9-94,135-149).

cv2 / torchvision are not available on the target image, so decoding / encoding go through PIL and the colour
swaps through numpy slicing.  Host-side semantics (dtype promotion, clamp, round-half-even) follow the reference;
the batched on-device versions are ops.img_u8_to_tensor / ops.tensor_to_img_u8.
"""
import math
import os

import numpy as np
import torch


def imread_bgr(path):
    """cv2.imread(path, cv2.IMREAD_COLOR) equivalent: uint8 HWC BGR (alpha dropped, gray replicated, EXIF orientation applied as
    cv2.imread does)."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        rgb = np.asarray(ImageOps.exif_transpose(im).convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def resize_bilinear(img, size, inv_scale=None):
    """cv2.resize(img, size, interpolation=cv2.INTER_LINEAR) for uint8 HWC images (identity when the size already matches).
    inv_scale=(fx, fy): the call form cv2.resize(img, (0, 0), fx=fx, fy=fy) -- OpenCV then maps coordinates with 1 / fx and 1 / fy
    instead of src / dst (they differ when fx * src is not an integer), `size` being (round(fx * w), round(fy * h)).

    cv2's rule, restated: half-pixel centres (src = (dst + 0.5) * scale - 0.5), the two nearest source samples per axis with NO
    antialiasing even when shrinking, source coordinates clamped to the image, and -- for uint8 -- fixed-point arithmetic: the
    per-axis weights are rounded to 11 bits (INTER_RESIZE_COEF_BITS, round-half-to-even like cvRound) and the 22-bit product sum
    is rounded to nearest with (v + 2^21) >> 22.  PIL's BILINEAR widens its support when shrinking, so it is not a stand-in."""
    h, w = img.shape[:2]
    dw, dh = int(size[0]), int(size[1])
    if (w, h) == (dw, dh):
        return img
    if img.dtype != np.uint8:
        raise TypeError('resize_bilinear restates the uint8 path of cv2.resize')

    def axis(n_src, n_dst, inv=None):
        scale = n_src / n_dst if inv is None else 1.0 / float(inv)
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0).astype(np.float32)
        lo = i0 < 0
        frac[lo] = 0.0
        i0[lo] = 0
        hi = i0 >= n_src - 1
        frac[hi] = 0.0
        i0[hi] = n_src - 1
        i1 = np.minimum(i0 + 1, n_src - 1)
        w1 = np.rint(frac.astype(np.float64) * 2048.0).astype(np.int64)     # cvRound: half to even
        return i0, i1, 2048 - w1, w1

    x0, x1, wx0, wx1 = axis(w, dw, inv_scale[0] if inv_scale else None)
    y0, y1, wy0, wy1 = axis(h, dh, inv_scale[1] if inv_scale else None)
    src = img.astype(np.int64).reshape(h, w, -1)
    rows0, rows1 = src[y0], src[y1]                                        # (dh, w, c)
    top = rows0[:, x0] * wx0[None, :, None] + rows0[:, x1] * wx1[None, :, None]
    bot = rows1[:, x0] * wx0[None, :, None] + rows1[:, x1] * wx1[None, :, None]
    out = (top * wy0[:, None, None] + bot * wy1[:, None, None] + (1 << 21)) >> 22
    return np.ascontiguousarray(out.astype(np.uint8).reshape((dh, dw) + img.shape[2:]))


def resize_linear_f32(img, size, inv_scale=None):
    """cv2.resize(..., interpolation=cv2.INTER_LINEAR) for float32 HWC images (the detector's `transform`, retinaface.py:160-161):
    float weights (1 - f, f), horizontal pass then vertical pass, half-pixel centres, clamped taps.  `inv_scale` = (fx, fy) when
    the call gives scale factors instead of a size: OpenCV then maps with 1 / fx rather than src / dst."""
    img = np.asarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    dw, dh = int(size[0]), int(size[1])

    def axis(n_src, n_dst, inv):
        scale = (1.0 / inv) if inv else n_src / n_dst
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        frac[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_src - 1
        frac[hi], i0[hi] = 0.0, n_src - 1
        return i0, np.minimum(i0 + 1, n_src - 1), (np.float32(1.0) - frac).astype(np.float32), frac

    x0, x1, ax0, ax1 = axis(w, dw, inv_scale[0] if inv_scale else None)
    y0, y1, ay0, ay1 = axis(h, dh, inv_scale[1] if inv_scale else None)
    src = img.reshape(h, w, -1)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]          # (h, dw, c) float32
    out = rows[y0] * ay0[:, None, None] + rows[y1] * ay1[:, None, None]
    return np.ascontiguousarray(out.astype(np.float32).reshape((dh, dw) + img.shape[2:]))


def resize_area(img, size):
    """cv2.resize(img, size, interpolation=cv2.INTER_AREA) for uint8 HWC images when shrinking on both axes (what
    FaceRestoreHelper.get_face_landmarks_5 feeds the detector for frames larger than `resize`, face_restoration_helper.py:208-215).

    OpenCV's rule, restated: every destination pixel is the mean of the source rectangle [d*s, (d+1)*s) with fractional end cells
    weighted by their covered length.  Integer ratios take the box-sum path (integer sums, one multiply by 1/area, round half to
    even; 2 x 2 boxes the integer form (sum + 2) >> 2); other ratios accumulate float32 products column weights first, then row weights, in source order, and round once."""
    if img.dtype != np.uint8:
        raise TypeError('resize_area restates the uint8 path of cv2.resize')
    h, w = img.shape[:2]
    dw, dh = int(size[0]), int(size[1])
    if (w, h) == (dw, dh):
        return img
    if dw > w or dh > h:
        raise ValueError('resize_area: INTER_AREA is restated for shrinking only (OpenCV switches to a linear rule when enlarging)')
    src = img.reshape(h, w, -1)
    c = src.shape[2]
    sx, sy = w / dw, h / dh
    if sx == int(sx) and sy == int(sy):
        kx, ky = int(sx), int(sy)
        box = np.zeros((dh, dw, c), dtype=np.uint32)
        for i in range(ky):
            for j in range(kx):
                box += src[i:dh * ky:ky, j:dw * kx:kx]
        if kx == 2 and ky == 2:          # OpenCV's 2x2 kernel is integer: (a + b + c + d + 2) >> 2, i.e. halves round UP
            return np.ascontiguousarray(((box + 2) >> 2).astype(np.uint8).reshape((dh, dw) + img.shape[2:]))
        out = np.rint(box.astype(np.float32) * np.float32(1.0 / (kx * ky)))
        return np.ascontiguousarray(np.clip(out, 0, 255).astype(np.uint8).reshape((dh, dw) + img.shape[2:]))

    def table(n_src, n_dst, scale):
        """(dst index, src index, weight) triples in OpenCV's order (computeResizeAreaTab)."""
        d = np.arange(n_dst)
        f1 = d * scale
        f2 = f1 + scale
        cell = np.minimum(scale, n_src - f1)
        s1 = np.ceil(f1).astype(np.int64)
        s2 = np.minimum(np.floor(f2).astype(np.int64), n_src - 1)
        s1 = np.minimum(s1, s2)
        di, si, al = [], [], []
        for k in range(n_dst):
            if s1[k] - f1[k] > 1e-3:
                di.append(k), si.append(s1[k] - 1), al.append(np.float32((s1[k] - f1[k]) / cell[k]))
            for t in range(s1[k], s2[k]):
                di.append(k), si.append(t), al.append(np.float32(1.0 / cell[k]))
            if f2[k] - s2[k] > 1e-3:
                di.append(k), si.append(s2[k]), al.append(np.float32(min(min(f2[k] - s2[k], 1.0), cell[k]) / cell[k]))
        return np.asarray(di), np.asarray(si), np.asarray(al, dtype=np.float32)

    xd, xs, xa = table(w, dw, sx)
    yd, ys, ya = table(h, dh, sy)
    try:                                  # both passes as sparse float32 products: a row's taps are added in table (= source) order
        from scipy import sparse
        wx = sparse.csr_matrix((xa, (xd, xs)), shape=(dw, w), dtype=np.float32)
        wy = sparse.csr_matrix((ya, (yd, ys)), shape=(dh, h), dtype=np.float32)
        cols = np.ascontiguousarray(src.transpose(1, 0, 2)).reshape(w, h * c).astype(np.float32)
        buf = np.asarray(wx @ cols, dtype=np.float32).reshape(dw, h, c)                 # buf[dx][sy] = sum_k src[sy][xs_k] * xa_k
        rows = np.ascontiguousarray(buf.transpose(1, 0, 2)).reshape(h, dw * c)
        out = np.asarray(wy @ rows, dtype=np.float32).reshape(dh, dw, c)                # sum_k buf[ys_k][dx] * ya_k
    except ImportError:
        srcf = src.astype(np.float32)
        buf = np.zeros((h, dw, c), dtype=np.float32)
        counts = np.bincount(xd, minlength=dw)
        starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
        for t in range(int(counts.max())):
            sel = np.nonzero(counts > t)[0]
            k = starts[sel] + t
            buf[:, sel] += srcf[:, xs[k]] * xa[k][None, :, None]
        out = np.zeros((dh, dw, c), dtype=np.float32)
        ycounts = np.bincount(yd, minlength=dh)
        ystarts = np.concatenate(([0], np.cumsum(ycounts)[:-1]))
        for t in range(int(ycounts.max())):
            sel = np.nonzero(ycounts > t)[0]
            k = ystarts[sel] + t
            out[sel] += buf[ys[k]] * ya[k][:, None, None]
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(out.reshape((dh, dw) + img.shape[2:]))


def img2tensor(imgs, bgr2rgb=True, float32=True):
    """HWC (BGR) ndarray(s) -> CHW (RGB) tensor(s); float64 input is narrowed to float32 before the swap."""

    def one(img):
        if img.ndim == 3 and img.shape[2] == 3 and bgr2rgb:
            if img.dtype == np.float64:
                img = img.astype(np.float32)
            img = img[:, :, ::-1]
        t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        return t.float() if float32 else t

    return [one(i) for i in imgs] if isinstance(imgs, list) else one(imgs)


def normalize_(t, mean, std):
    """In-place (t - mean) / std per channel (torchvision.transforms.functional.normalize, inplace=True)."""
    m = torch.as_tensor(mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
    return t.sub_(m).div_(s)


def _grid(t, nrow, padding=2):
    """Minimal torchvision.utils.make_grid(normalize=False) for (B,C,H,W)."""
    b, c, h, w = t.shape
    if c == 1:
        t = t.expand(b, 3, h, w)
        c = 3
    xmaps = min(nrow, b)
    ymaps = int(math.ceil(b / xmaps))
    H, W = h + padding, w + padding
    grid = t.new_zeros((c, H * ymaps + padding, W * xmaps + padding))
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= b:
                break
            grid[:, y * H + padding:y * H + padding + h, x * W + padding:x * W + padding + w] = t[k]
            k += 1
    return grid


def tensor2img(tensor, rgb2bgr=True, out_type=np.uint8, min_max=(0, 1)):
    """Tensor(s) (B,3|1,H,W) / (3|1,H,W) / (H,W), RGB -> ndarray(s) HWC BGR.

    clamp to min_max, rescale to [0,1]; uint8 output = round-half-even(x*255) (basicsr/utils/img_util.py:66-90).
    """
    if not (torch.is_tensor(tensor) or (isinstance(tensor, list) and all(torch.is_tensor(t) for t in tensor))):
        raise TypeError(f'tensor or list of tensors expected, got {type(tensor)}')
    tensors = [tensor] if torch.is_tensor(tensor) else tensor
    result = []
    for t in tensors:
        t = t.squeeze(0).float().detach().cpu().clamp_(*min_max)
        t = (t - min_max[0]) / (min_max[1] - min_max[0])
        if t.dim() == 4:
            t = _grid(t, nrow=int(math.sqrt(t.size(0))))
        if t.dim() == 3:
            img = t.numpy().transpose(1, 2, 0)
            if img.shape[2] == 1:
                img = np.squeeze(img, axis=2)
            elif rgb2bgr:
                img = img[:, :, ::-1]
        elif t.dim() == 2:
            img = t.numpy()
        else:
            raise TypeError(f'Only support 4D, 3D or 2D tensor. But received with dimension: {t.dim()}')
        if out_type == np.uint8:
            img = (img * 255.0).round()
        result.append(np.ascontiguousarray(img.astype(out_type)))
    return result[0] if len(result) == 1 else result


def imwrite(img, file_path, params=None, auto_mkdir=True, compress_level=None):
    """Write a uint8 HWC BGR (or HW gray) array; format from the extension (cv2.imwrite stand-in).  compress_level: zlib level
    for PNG output (None = PIL's default 6; decoded pixels are identical at every level)."""
    from PIL import Image
    if auto_mkdir:
        os.makedirs(os.path.abspath(os.path.dirname(file_path)), exist_ok=True)
    arr = np.asarray(img)
    if arr.ndim == 3:
        arr = arr[:, :, ::-1]
    kw = {'compress_level': int(compress_level)} if (compress_level is not None and str(file_path).lower().endswith('.png')) else {}
    Image.fromarray(np.ascontiguousarray(arr)).save(file_path, **kw)
    return True
