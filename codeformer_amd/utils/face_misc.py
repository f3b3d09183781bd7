"""Host-side helpers the aligned-face entrypoint takes from the reference's facelib (facelib/utils/misc.py:146-201,
facelib/utils/face_restoration_helper.py:364-369): gray-input test and the numpy colour transfer applied to restored
gray faces.  uint8 HWC BGR arrays in, same out."""
import numpy as np


def is_gray(img, threshold=10):
    """True when the three channels are (nearly) identical: mean variance of the pairwise channel differences."""
    if img.ndim == 2 or img.shape[2] == 1:
        return True
    c = [img[:, :, i].astype(np.int16) for i in range(3)]
    diff = ((c[0] - c[1]).var() + (c[1] - c[2]).var() + (c[2] - c[0]).var()) / 3.0
    return bool(diff <= threshold)


def bgr2gray(img, out_channel=3):
    b, g, r = img[:, :, 0], img[:, :, 1], img[:, :, 2]
    gray = 0.2989 * r + 0.5870 * g + 0.1140 * b
    return gray[:, :, np.newaxis].repeat(3, axis=2) if out_channel == 3 else gray


def _mean_std(feat, eps=1e-5):
    c = feat.shape[2]
    flat = feat.reshape(-1, c)
    return flat.mean(axis=0).reshape(1, 1, c), np.sqrt(flat.var(axis=0) + eps).reshape(1, 1, c)


def adain_npy(content_feat, style_feat):
    s_mean, s_std = _mean_std(style_feat)
    c_mean, c_std = _mean_std(content_feat)
    return (content_feat - c_mean) / c_std * s_std + s_mean


class AlignedFaceHelper:
    """The slice of FaceRestoreHelper the --has_aligned path touches (clean_all / cropped_faces / add_restored_face /
    restored_faces; inference_codeformer.py:167,183-186,214,232) -- without constructing the detector and parser nets
    the reference builds (and downloads) even for aligned inputs."""

    def __init__(self):
        self.is_gray = False
        self.cropped_faces = []
        self.restored_faces = []

    def clean_all(self):
        self.cropped_faces = []
        self.restored_faces = []

    def add_restored_face(self, restored_face, input_face=None):
        if self.is_gray:
            restored_face = bgr2gray(restored_face)
            if input_face is not None:
                restored_face = adain_npy(restored_face, input_face)
        self.restored_faces.append(restored_face)
