"""`RealESRGANer`: image-level driver around RRDBNet -- pad, (optionally) tile, run, crop
(mirror of basicsr/utils/realesrgan_utils.py:14-228; built at inference_codeformer.py:37-45).

Same constructor arguments and `enhance(img, outscale, alpha_upsampler) -> (output, img_mode)` contract: `img` is an
HxWx{1,3,4} BGR(A) uint8 / uint16 array as cv2.imread returns it, the result is BGR(A) in the same integer range.
The network runs on the HIP kernels in fp32, or -- `half=True`, the reference's GPU default -- with IEEE-half MFMA operands
(fp32 accumulation, fp32 tensors: `RRDBNet.half()`); colour handling is plain numpy (cv2 is not a dependency).
Tiling is kept for API parity (`tile` > 0) but a 288 GB device does not need it: `tile=0` runs the whole image at once.
"""
import os

import numpy as np
import torch
from torch.nn import functional as F

from .download_util import load_file_from_url
from .misc import get_device


def _spans(length, tile, pad):
    """Split [0, length) into tiles: yields (core_lo, core_hi, padded_lo, padded_hi) per tile."""
    lo = 0
    while lo < length:
        hi = min(lo + tile, length)
        yield lo, hi, max(lo - pad, 0), min(hi + pad, length)
        lo = hi


class RealESRGANer:

    def __init__(self, scale, model_path, model=None, tile=0, tile_pad=10, pre_pad=10, half=False, device=None, gpu_id=None):
        self.scale, self.tile_size, self.tile_pad, self.pre_pad = scale, tile, tile_pad, pre_pad
        self.mod_scale = {2: 2, 1: 4}.get(scale)   # the pixel-unshuffle factor the input size must be divisible by
        self.half = bool(half)   # -> RRDBNet.precision 'fp16' (f16 MFMA operands); tensors handed to / from the model stay fp32
        self.device = get_device(gpu_id) if device is None else torch.device(device)
        if model_path is not None:
            if model_path.startswith('https://'):
                model_path = load_file_from_url(url=model_path, model_dir=os.path.join('weights/realesrgan'), progress=True,
                                                file_name=None)
            ckpt = torch.load(model_path, map_location='cpu')
            model.load_state_dict(ckpt['params_ema' if 'params_ema' in ckpt else 'params'], strict=True)
        self.model = model.eval().to(self.device)
        if self.half and self.device.type == 'cuda':
            self.model = self.model.half()

    # -- tensor stages ---------------------------------------------------------------------------------------------------
    def pre_process(self, img):
        """HxWx3 float array -> self.img (1,3,H',W'): reflect pre-pad (right/bottom), then reflect-pad to a multiple of
        the unshuffle factor (realesrgan_utils.py:70-93)."""
        t = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)))).float().unsqueeze(0).to(self.device)
        if self.pre_pad:
            t = F.pad(t, (0, self.pre_pad, 0, self.pre_pad), 'reflect')
        self.mod_pad_h = self.mod_pad_w = 0
        if self.mod_scale:
            h, w = t.shape[2:]
            self.mod_pad_h, self.mod_pad_w = (-h) % self.mod_scale, (-w) % self.mod_scale
            t = F.pad(t, (0, self.mod_pad_w, 0, self.mod_pad_h), 'reflect')
        self.img = t

    def process(self):
        self.output = self.model(self.img)

    def tile_process(self):
        """Run padded tiles and keep each tile's un-padded core (realesrgan_utils.py:99-163)."""
        b, c, h, w = self.img.shape
        s = self.scale
        self.output = self.img.new_zeros((b, c, h * s, w * s))
        for y0, y1, py0, py1 in _spans(h, self.tile_size, self.tile_pad):
            for x0, x1, px0, px1 in _spans(w, self.tile_size, self.tile_pad):
                up = self.model(self.img[:, :, py0:py1, px0:px1])
                oy, ox = (y0 - py0) * s, (x0 - px0) * s
                self.output[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = up[:, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]

    def post_process(self):
        _, _, h, w = self.output.shape
        cut_h, cut_w = (self.mod_pad_h + self.pre_pad) * self.scale, (self.mod_pad_w + self.pre_pad) * self.scale
        self.output = self.output[:, :, :h - cut_h, :w - cut_w]
        return self.output

    def _upscale(self, rgb):
        self.pre_process(rgb)
        self.tile_process() if self.tile_size > 0 else self.process()
        out = self.post_process().squeeze(0).float().clamp_(0, 1).cpu().numpy()
        return np.transpose(out[::-1], (1, 2, 0))   # RGB planes -> HxWx3 BGR

    # -- image level -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def enhance(self, img, outscale=None, alpha_upsampler='realesrgan'):
        max_range = 65535 if np.max(img) > 256 else 255
        x = img.astype(np.float32) / max_range
        alpha = None
        if x.ndim == 2:
            img_mode, rgb = 'L', np.repeat(x[:, :, None], 3, axis=2)
        elif x.shape[2] == 4:
            img_mode, rgb, alpha = 'RGBA', x[:, :, 2::-1], x[:, :, 3]
        else:
            img_mode, rgb = 'RGB', x[:, :, ::-1]
        out = self._upscale(rgb)
        if img_mode == 'L':   # BGR -> gray with the BT.601 weights cv2.COLOR_BGR2GRAY uses
            out = out[:, :, 0] * 0.114 + out[:, :, 1] * 0.587 + out[:, :, 2] * 0.299
        if alpha is not None:
            if alpha_upsampler != 'realesrgan':
                raise NotImplementedError("alpha_upsampler other than 'realesrgan' needs cv2.resize")
            a = self._upscale(np.repeat(alpha[:, :, None], 3, axis=2))
            a = a[:, :, 0] * 0.114 + a[:, :, 1] * 0.587 + a[:, :, 2] * 0.299
            out = np.concatenate([out, a[:, :, None]], axis=2)
        out = np.round(out * float(max_range)).astype(np.uint16 if max_range == 65535 else np.uint8)
        if outscale is not None and float(outscale) != float(self.scale):
            raise NotImplementedError('outscale != network scale needs a LANCZOS4 resize (cv2); resize on the caller side')
        return out, img_mode
