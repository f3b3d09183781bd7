"""Device-side alignment warp and paste-back: the detection-independent half of the reference's FaceRestoreHelper.

Mirror of facelib/utils/face_restoration_helper.py -- `align_warp_face` (:320-350), `get_inverse_affine` (:352-362),
`add_restored_face` (:364-369) and `paste_faces_to_input_image` (:372-499; square soft mask, optional parse-mask fusion) -- with the
frame, the crops, the masks and the upsampled canvas living in HBM and every OpenCV call replaced by a kernel of cf_paste.hip over the
face's bounding box (the reference filters the whole upsampled frame once per face).  Face DETECTION (RetinaFace, landmarks,
`cv2.estimateAffinePartial2D`) stays on the host as the north star says: the 2x3 alignment matrices are this class's input.

Differences from the reference, all stated in DESIGN.md: one device->host read-back per FRAME (the face areas that size the
feathering kernel, :433-441) instead of arrays crossing PCIe at every step; the parse-mask branch blends in float32 where numpy
promotes to float64 (the final image is truncated to uint8, so at most rare 1-LSB differences); `draw_box` and the alpha channel of
RGBA backgrounds are not built.
"""
import math

import numpy as np
import torch

from .. import ops

MASK_COLORMAP = (0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 0, 255, 0, 0, 0)   # :467


def invert_affine(m):
    """cv2.invertAffineTransform (double precision)."""
    m = np.asarray(m, dtype=np.float64).reshape(2, 3)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * d, m[0, 0] * d, -m[0, 1] * d, -m[1, 0] * d
    return np.array([[a11, a12, -a11 * m[0, 2] - a12 * m[1, 2]], [a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]]], dtype=np.float64)


def gaussian_taps(ksize, sigma=0.0):
    """cv2.getGaussianKernel(ksize, sigma) as float32 taps: sigma <= 0 -> 0.3*((ksize-1)*0.5 - 1) + 0.8, with OpenCV's fixed tables
    for ksize <= 7 in that case."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in small:
        return np.asarray(small[ksize], dtype=np.float32)
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


class DeviceFaceHelper:
    """Per-frame state on the device: read_image -> align_warp_face(affines) -> [restore the crops] -> add_restored_faces ->
    paste_faces_to_input_image().  Method names follow FaceRestoreHelper; arrays are uint8 HWC BGR like cv2's."""

    def __init__(self, upscale_factor=1, face_size=512, device='cuda', use_parse=False, face_parse=None):
        self.upscale_factor = int(upscale_factor)
        self.face_size = (int(face_size), int(face_size))
        self.device = torch.device(device)
        self.use_parse = bool(use_parse)
        self.face_parse = face_parse              # a codeformer_amd.facelib.parsing ParseNet on `device` (needed when use_parse)
        self._ones = torch.ones(self.face_size[1], self.face_size[0], dtype=torch.float32, device=self.device)
        self._taps = {}
        self._partials = None
        self.clean_all()

    def clean_all(self):
        self.input_img = None
        self.affine_matrices = []
        self.inverse_affine_matrices = []
        self.cropped_faces = None
        self.restored_faces = None

    def read_image(self, img):
        """img: uint8 (H,W,3) BGR numpy array or CUDA tensor (the decoded frame)."""
        t = torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError('read_image expects a uint8 HxWx3 image')
        self.input_img = t.to(self.device, non_blocking=True).contiguous()

    # ---- align_warp_face (:320-350) ------------------------------------------------------------------------------------------------
    def align_warp_face(self, affine_matrices):
        """affine_matrices: (n,2,3) frame -> face matrices (what estimateAffinePartial2D returned on the host).  All n crops are cut in
        ONE launch: cv2.warpAffine(input_img, M, face_size, borderMode=BORDER_CONSTANT, borderValue=(135, 133, 132)).
        Returns the uint8 (n, 512, 512, 3) crops on the device."""
        aff = np.asarray(affine_matrices, dtype=np.float64).reshape(-1, 2, 3)
        self.affine_matrices = [a.copy() for a in aff]
        n = aff.shape[0]
        fw, fh = self.face_size
        crops = torch.empty(n, fh, fw, 3, dtype=torch.uint8, device=self.device)
        if n:
            inv = np.stack([invert_affine(a) for a in aff]).reshape(n, 6)           # destination (face) pixel -> frame coordinate
            ops.warp_affine_u8(self.input_img, torch.from_numpy(inv).to(self.device), crops, border=(135, 133, 132))
        self.cropped_faces = crops
        return crops

    def get_inverse_affine(self):
        self.inverse_affine_matrices = [invert_affine(a) * self.upscale_factor for a in self.affine_matrices]   # :354-356
        return self.inverse_affine_matrices

    def add_restored_faces(self, restored):
        """restored: uint8 (n,512,512,3) BGR CUDA tensor (cf_tensor_to_img_u8 output)."""
        self.restored_faces = restored

    # ---- parse mask (:455-481) -------------------------------------------------------------------------------------------------------
    def parse_soft_masks(self, restored):
        """ParseNet labels -> 0/255 map -> GaussianBlur(101, 11) twice -> 10-pixel border cleared -> / 255, per face, on the device."""
        if self.face_parse is None:
            raise RuntimeError('use_parse needs a ParseNet (face_parse=...)')
        x = ops.img_u8_to_tensor(restored)                                            # img2tensor(face/255.) + normalize(0.5, 0.5)
        labels = self.face_parse.parse_labels(x)                                      # (n,512,512) int64
        m = ops.label_lut(labels, MASK_COLORMAP)
        taps = self._taps_dev(101, 11.0)
        out = torch.empty_like(m)
        for i in range(m.shape[0]):
            out[i] = ops.gaussian_blur(ops.gaussian_blur(m[i], taps), taps)
        return ops.scale_clear_border_(out, 10, 1.0 / 255.0)

    def _taps_dev(self, ksize, sigma=0.0):
        key = (ksize, sigma)
        if key not in self._taps:
            self._taps[key] = torch.from_numpy(gaussian_taps(ksize, sigma)).to(self.device)
        return self._taps[key]

    # ---- paste_faces_to_input_image (:372-499) -----------------------------------------------------------------------------------------
    def _region(self, inv_up, h_up, w_up):
        """Bounding box of the warped face square, grown by everything the masks can spread (erosions only shrink; the Gaussian
        spreads by its radius) and clipped to the canvas: outside it the soft mask is exactly 0."""
        fw, fh = self.face_size
        corners = np.array([[0, 0, 1], [fw, 0, 1], [0, fh, 1], [fw, fh, 1]], dtype=np.float64).T
        q = inv_up @ corners
        area = abs(inv_up[0, 0] * inv_up[1, 1] - inv_up[0, 1] * inv_up[1, 0]) * fw * fh
        margin = int(math.sqrt(area)) // 20 + 8
        x0, x1 = int(math.floor(q[0].min())) - margin, int(math.ceil(q[0].max())) + margin + 1
        y0, y1 = int(math.floor(q[1].min())) - margin, int(math.ceil(q[1].max())) + margin + 1
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, w_up), min(y1, h_up)
        return (x0, y0, x1 - x0, y1 - y0) if (x1 > x0 and y1 > y0) else None

    @torch.no_grad()
    def paste_faces_to_input_image(self, upsample_img=None, return_tensor=False):
        """Returns the pasted uint8 (h_up, w_up, 3) BGR frame (numpy, or the CUDA tensor with return_tensor=True).
        upsample_img: optional uint8 background already at (h_up, w_up) (e.g. the Real-ESRGAN output)."""
        h, w, _ = self.input_img.shape
        u = self.upscale_factor
        h_up, w_up = int(h * u), int(w * u)
        if upsample_img is None:
            canvas = ops.resize_linear_u8(self.input_img, h_up, w_up)                # cv2.resize(input_img, (w_up, h_up), INTER_LINEAR), :380
        else:
            bg = torch.from_numpy(np.ascontiguousarray(upsample_img)).to(self.device) if isinstance(upsample_img, np.ndarray) else upsample_img
            if tuple(bg.shape) != (h_up, w_up, 3):
                raise ValueError(f'upsample_img must be {(h_up, w_up, 3)} (resize it first), got {tuple(bg.shape)}')
            canvas = ops.resize_linear_u8(bg.contiguous(), h_up, w_up)
        faces = self.restored_faces
        n = 0 if faces is None else faces.shape[0]
        if not self.inverse_affine_matrices:
            self.get_inverse_affine()
        assert n == len(self.inverse_affine_matrices), 'length of restored_faces and affine_matrices are different.'
        parse = self.parse_soft_masks(faces) if (self.use_parse and n) else None
        # phase 1 (all faces): warped square mask, first erosion, area partials -> ONE read-back per frame
        if self._partials is None or self._partials.shape[0] < max(n, 1):
            self._partials = torch.empty(max(n, 1), 64, dtype=torch.float64, device=self.device)
        work = []
        for k in range(n):
            inv = self.inverse_affine_matrices[k].copy()
            inv[:, 2] += 0.5 * u if u > 1 else 0                                       # :393-398
            region = self._region(inv, h_up, w_up)
            if region is None:
                work.append(None)
                self._partials[k].zero_()
                continue
            dst2src = invert_affine(inv)                                               # what cv2.warpAffine derives from its M
            inv_mask = ops.warp_affine_f32(self._ones, dst2src, region)                # :428
            ero = ops.erode(inv_mask, int(2 * u))                                      # :430-431
            ops.sum_partials(ero, self._partials[k])                                   # :433
            work.append((region, dst2src, ero))
        areas = self._partials[:n].sum(dim=1).cpu().numpy() if n else []
        # phase 2: feathered mask + blend, face by face (the order matters where faces overlap)
        for k in range(n):
            if work[k] is None:
                continue
            region, dst2src, ero = work[k]
            w_edge = int(float(areas[k]) ** 0.5) // 20                                 # :441
            center = ops.erode(ero, w_edge * 2)                                        # :443
            soft = ops.gaussian_blur(center, self._taps_dev(w_edge * 2 + 1), (region[0], region[1]), (h_up, w_up))   # :445
            pm = ops.warp_affine_f32(parse[k], dst2src, region) if parse is not None else None   # :478-479
            ops.paste_blend(canvas, faces[k], dst2src, ero, soft, region, parse=pm)    # :400, :432, :481-492
        out = ops.f32_to_u8_trunc(canvas)                                              # :497
        return out if return_tensor else out.cpu().numpy()
