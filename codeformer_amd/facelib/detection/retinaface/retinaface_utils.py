"""Anchor generation, box / landmark decoding and NMS of the RetinaFace detector (inference half of the reference's
facelib/detection/retinaface/retinaface_utils.py: PriorBox :8-36, py_cpu_nms :39-47, decode :254-271, decode_landm :274-294,
batched_decode :297-317, batched_decode_landm :320-340).  The training half (match / encode / jaccard ...) is out of scope.

`py_cpu_nms` in the reference is torchvision.ops.nms (absent here): greedy suppression in descending score order, IoU
= inter / (area_i + area_j - inter) in float32 with no +1 on the extents, a box is dropped when IoU > thresh.
"""
from math import ceil

import numpy as np
import torch


class PriorBox(object):
    """Anchors (cx, cy, w, h) in image-relative units: for every pyramid level k (stride steps[k]) and every cell (i, j) of its
    ceil(H / step) x ceil(W / step) grid, one anchor per entry of min_sizes[k], centred at ((j + .5) step / W, (i + .5) step / H).
    Evaluated in float64 like the reference's Python floats, then stored as float32."""

    def __init__(self, cfg, image_size=None, phase='train'):
        self.min_sizes, self.steps, self.clip = cfg['min_sizes'], cfg['steps'], cfg['clip']
        self.image_size = image_size
        self.feature_maps = [[ceil(image_size[0] / s), ceil(image_size[1] / s)] for s in self.steps]
        self.name = 's'

    def forward(self):
        H, W = float(self.image_size[0]), float(self.image_size[1])
        levels = []
        for (fh, fw), step, sizes in zip(self.feature_maps, self.steps, self.min_sizes):
            cy = (np.arange(fh, dtype=np.float64) + 0.5) * step / H
            cx = (np.arange(fw, dtype=np.float64) + 0.5) * step / W
            a = np.empty((fh, fw, len(sizes), 4), dtype=np.float64)
            a[..., 0] = cx[None, :, None]
            a[..., 1] = cy[:, None, None]
            a[..., 2] = np.asarray(sizes, dtype=np.float64)[None, None, :] / W
            a[..., 3] = np.asarray(sizes, dtype=np.float64)[None, None, :] / H
            levels.append(a.reshape(-1, 4))
        out = torch.from_numpy(np.concatenate(levels, 0).astype(np.float32))
        return out.clamp_(min=0, max=1) if self.clip else out


def py_cpu_nms(dets, thresh):
    """dets: (n, 5) [x1, y1, x2, y2, score]; returns the kept row indices, highest score first."""
    dets = np.asarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    area = (x2 - x1) * (y2 - y1)
    order = np.argsort(-dets[:, 4], kind='stable')
    alive = np.ones(order.size, dtype=bool)
    keep = []
    thresh = np.float32(thresh)
    for pos, i in enumerate(order):
        if not alive[pos]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = inter / (area[i] + area[rest] - inter)
        alive[pos + 1:] &= ~(iou > thresh)
    return keep


def _corners(centre_size):
    """(..., [cx, cy, w, h]) -> (..., [x1, y1, x2, y2]) with the reference's in-place order: x1 = cx - w/2, then x2 = w + x1."""
    lo = centre_size[..., :2] - centre_size[..., 2:] / 2
    return torch.cat((lo, centre_size[..., 2:] + lo), dim=-1)


def _decode_any(loc, priors, variances):
    centre = priors[..., :2] + loc[..., :2] * variances[0] * priors[..., 2:]
    size = priors[..., 2:] * torch.exp(loc[..., 2:] * variances[1])
    return _corners(torch.cat((centre, size), dim=-1))


def _decode_landm_any(pre, priors, variances):
    pts = [priors[..., :2] + pre[..., 2 * k:2 * k + 2] * variances[0] * priors[..., 2:] for k in range(5)]
    return torch.cat(pts, dim=-1)


def decode(loc, priors, variances):
    """loc (n,4) regression output, priors (n,4) -> boxes (n,4) corner form, image-relative units."""
    return _decode_any(loc, priors, variances)


def decode_landm(pre, priors, variances):
    """pre (n,10) -> five (x, y) landmarks per prior, image-relative units."""
    return _decode_landm_any(pre, priors, variances)


def batched_decode(b_loc, priors, variances):
    """b_loc (B,n,4), priors (1,n,4)."""
    return _decode_any(b_loc, priors, variances)


def batched_decode_landm(pre, priors, variances):
    """pre (B,n,10), priors (1,n,4)."""
    return _decode_landm_any(pre, priors, variances)
