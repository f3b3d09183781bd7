"""RetinaFace detection + alignment on the host (last clause; BASELINE config 4: "facelib detect/align on host").

CPU: this repo's detector (state_dict keys, forward, anchors, decode, NMS, detect_faces / batched_detect_faces) against goldens
produced by the REFERENCE's own retinaface files (oracle/make_golden_retinaface.py; torchvision's ResNet-50 / nms restated in
oracle/tv_stub.py), the LMedS similarity fit (restated from OpenCV, parity unpinned) against its defining properties, and the
INTER_AREA / float INTER_LINEAR resizes against direct definitions.
Tolerances (fp32 CPU, same ATen calls in the same order): network outputs 2e-5 on values up to 3; boxes 1e-3 pixel.
"""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _digests():
    with open(os.path.join(GOLD, 'retinaface_digests.json')) as f:
        return json.load(f)


def _build(case, **kw):
    from facelib.detection.retinaface.retinaface import RetinaFace
    from oracle.make_golden_retinaface import build
    return build(RetinaFace, case, **kw)


def _frames(case):
    from oracle.make_golden_retinaface import seeded_frames
    return seeded_frames(tuple(case['shape']), case['in_seed'])


@pytest.mark.parametrize('name', ['resnet50', 'mobile0.25'])
def test_detector_matches_reference_goldens(name):
    from oracle.make_golden_retinaface import synth_keys_sha
    case = _digests()[name]
    gold = np.load(os.path.join(GOLD, f"retinaface_{name.replace('.', '')}.npz"))
    net = _build(case)
    assert len(net.state_dict()) == case['n_keys'] and synth_keys_sha(net.state_dict()) == case['keys_sha'], \
        'state_dict keys / shapes / order differ from the reference module'
    frames = _frames(case)
    x = torch.from_numpy(frames.astype(np.float32)).permute(0, 3, 1, 2) - net.mean_tensor
    with torch.no_grad():
        loc, conf, ldm = net(x)
    for got, key in ((loc, 'loc'), (conf, 'conf'), (ldm, 'ldm')):
        err = float((got - torch.from_numpy(gold[key])).abs().max())
        assert err <= 2e-5, (key, err)
    # anchors: bit-equal to the reference's Python-float loop
    from facelib.detection.retinaface.retinaface_utils import PriorBox
    pri = PriorBox(net.cfg, image_size=(case['shape'][1], case['shape'][2])).forward()
    assert pri.shape[0] == case['n_priors'] and np.array_equal(pri.numpy(), gold['priors'])
    # single-image and batched flows
    det = net.detect_faces(frames[0], conf_threshold=case['conf'], nms_threshold=case['nms'])
    assert det.shape == gold['det'].shape == (case['n_det'], 15) and det.dtype == np.float32
    assert float(np.abs(det - gold['det']).max()) <= 1e-3
    bb, bl = net.batched_detect_faces(torch.from_numpy(frames.astype(np.float32)), conf_threshold=case['conf'],
                                      nms_threshold=case['nms'])
    assert [b.shape[0] for b in bb] == case['n_batched']
    for i in range(len(bb)):
        assert float(np.abs(bb[i] - gold[f'bbox{i}']).max()) <= 1e-3 and float(np.abs(bl[i] - gold[f'bldm{i}']).max()) <= 1e-3
    # a frame without detections gives empty arrays
    bb0, bl0 = net.batched_detect_faces(torch.from_numpy(frames[:1].astype(np.float32)), conf_threshold=2.0)
    assert bb0[0].size == 0 and bl0[0].size == 0
    assert net.detect_faces(frames[0], conf_threshold=2.0).shape == (0, 15)


def test_nms_against_definition():
    from facelib.detection.retinaface.retinaface_utils import py_cpu_nms
    from oracle import tv_stub
    rs = np.random.RandomState(3)
    xy = rs.rand(200, 2) * 100
    wh = rs.rand(200, 2) * 40 + 5
    dets = np.hstack([xy, xy + wh, rs.rand(200, 1)]).astype(np.float32)
    keep = py_cpu_nms(dets, 0.4)
    ref = tv_stub.nms(torch.from_numpy(dets[:, :4]), torch.from_numpy(dets[:, 4]), 0.4).tolist()
    assert keep == ref and 10 < len(keep) < 200
    assert py_cpu_nms(np.zeros((0, 5), np.float32), 0.4) == []


def test_init_detection_model_and_shim(tmp_path):
    import facelib.detection as fd
    from facelib.detection.retinaface.retinaface import RetinaFace
    with pytest.raises(NotImplementedError):
        fd.init_detection_model('YOLOv5l')
    with pytest.raises(NotImplementedError):
        RetinaFace('resnet18')
    with pytest.raises(FileNotFoundError):
        fd.init_detection_model('retinaface_mobile0.25')
    ref = RetinaFace('mobile0.25')
    path = tmp_path / 'detection_mobilenet0.25_Final.pth'
    torch.save({'module.' + k: v for k, v in ref.state_dict().items()}, path)        # as saved from DataParallel
    net = fd.init_detection_model('retinaface_mobile0.25', device='cpu', model_path=str(path))
    assert not net.training and all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))


# ---------------------------------------------------------------------------------------------------------------- alignment fit

def _template():
    from codeformer_amd.facelib.utils.face_restoration_helper import _TEMPLATE_5
    return np.array(_TEMPLATE_5, dtype=np.float64)


def _landmarks(theta, s, t, noise=0.0, seed=0):
    tpl = _template()
    R = s * np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    src = (tpl - t) @ np.linalg.inv(R).T
    return src + np.random.RandomState(seed).randn(5, 2) * noise, np.hstack([R, np.asarray(t, dtype=np.float64).reshape(2, 1)])


def test_lmeds_similarity_fit():
    from codeformer_amd.facelib.align import _Rng, estimate_affine_partial_2d, least_squares_similarity
    tpl = _template()
    # exact similarity data: recovered to float32 accuracy, all inliers
    src, M = _landmarks(0.3, 2.5, (40.0, -12.0))
    got, inl = estimate_affine_partial_2d(src, tpl)
    assert inl.ravel().tolist() == [1] * 5 and np.abs(got - M).max() < 2e-3 * np.abs(M).max()
    assert abs(got[0, 0] - got[1, 1]) == 0 and abs(got[0, 1] + got[1, 0]) == 0          # 4-dof form
    # noisy landmarks without outliers: the least-squares similarity of all five points (on the float32-converted inputs)
    src, _ = _landmarks(-0.2, 1.7, (10.0, 30.0), noise=1.0, seed=4)
    got, inl = estimate_affine_partial_2d(src, tpl)
    ls = least_squares_similarity(src.astype(np.float32), tpl.astype(np.float32))
    assert inl.sum() == 5 and np.allclose(got, ls, rtol=0, atol=1e-12)
    # the least-squares fit is a stationary point of the reprojection error
    def cost(m):
        return float((((src.astype(np.float32).astype(np.float64) @ m[:, :2].T + m[:, 2]) - tpl.astype(np.float32)) ** 2).sum())
    base = cost(ls)
    for da, db in ((1e-4, 0), (-1e-4, 0), (0, 1e-4), (0, -1e-4)):
        m = ls.copy()
        m[0, 0] += da; m[1, 1] += da; m[0, 1] -= db; m[1, 0] += db
        assert cost(m) >= base
    # one gross outlier is rejected and does not move the fit
    bad = src.copy()
    bad[2] += (80.0, -60.0)
    got_o, inl_o = estimate_affine_partial_2d(bad, tpl)
    assert inl_o.ravel().tolist() == [1, 1, 0, 1, 1]
    keep = [0, 1, 3, 4]
    assert np.allclose(got_o, least_squares_similarity(src[keep].astype(np.float32), tpl[keep].astype(np.float32)), atol=1e-12)
    # deterministic (OpenCV seeds its generator per call) and degenerate inputs
    again, _ = estimate_affine_partial_2d(bad, tpl)
    assert np.array_equal(again, got_o)
    assert estimate_affine_partial_2d(src[:1], tpl[:1])[0] is None
    two, _ = estimate_affine_partial_2d(src[:2], tpl[:2])
    assert np.abs(src[:2].astype(np.float32) @ two[:, :2].T + two[:, 2] - tpl[:2].astype(np.float32)).max() < 1e-3
    # the generator: multiply-with-carry, first values from the all-ones state
    r = _Rng()
    s = 0xFFFFFFFFFFFFFFFF
    for _ in range(3):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert r.uniform(0, 5) == (s & 0xFFFFFFFF) % 5


def test_helper_selection_rules():
    from facelib.utils.face_restoration_helper import get_center_face, get_largest_face
    dets = [np.array([10, 10, 60, 60, 0.9]), np.array([100, 100, 300, 320, 0.8]), np.array([380, 10, 470, 90, 0.95])]
    assert get_largest_face(dets, 400, 500)[1] == 1
    assert get_center_face(dets, 400, 500)[1] == 1
    assert get_center_face(dets, center=(30, 30))[1] == 0
    assert get_largest_face([np.array([-50, -50, 40, 40, 1.0]), np.array([0, 0, 45, 45, 1.0])], 100, 100)[1] == 1   # clipped to the image


# ---------------------------------------------------------------------------------------------------------------- resizes

def test_resize_area_and_float_linear():
    from basicsr.utils.img_util import resize_area, resize_linear_f32
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (48, 60, 3)).astype(np.uint8)
    # integer ratio: plain box mean, rounded half to even
    out = resize_area(img, (20, 16))
    box = img.reshape(16, 3, 20, 3, 3).astype(np.float64).mean(axis=(1, 3))
    assert out.shape == (16, 20, 3) and np.abs(out.astype(np.float64) - box).max() <= 0.5 + 1e-6
    half = resize_area(img, (30, 24))                                    # 2 x 2 boxes: (sum + 2) >> 2
    assert np.array_equal(half, ((img.reshape(24, 2, 30, 2, 3).astype(np.int64).sum(axis=(1, 3)) + 2) >> 2).astype(np.uint8))
    # fractional ratio: area-weighted mean of the covered source rectangle
    out = resize_area(img, (25, 20))
    sx, sy = 60 / 25, 48 / 20
    ref = np.zeros((20, 25, 3))
    for dy in range(20):
        for dx in range(25):
            acc, wsum = np.zeros(3), 0.0
            for y in range(int(np.floor(dy * sy)), int(np.ceil((dy + 1) * sy))):
                wy = min(y + 1, (dy + 1) * sy) - max(y, dy * sy)
                for x in range(int(np.floor(dx * sx)), int(np.ceil((dx + 1) * sx))):
                    wx = min(x + 1, (dx + 1) * sx) - max(x, dx * sx)
                    acc += wx * wy * img[y, x]
                    wsum += wx * wy
            ref[dy, dx] = acc / wsum
    assert np.abs(out.astype(np.float64) - ref).max() <= 0.5 + 1e-3
    assert resize_area(img, (60, 48)) is img
    with pytest.raises(ValueError):
        resize_area(img, (61, 48))
    # float bilinear: half-pixel centres, exact on a linear ramp away from the clamped border
    yy, xx = np.mgrid[0:30, 0:40].astype(np.float32)
    ramp = np.stack([xx, yy, xx + yy], axis=2)
    up = resize_linear_f32(ramp, (80, 60))
    cx = (np.arange(80) + 0.5) / 2 - 0.5
    cy = (np.arange(60) + 0.5) / 2 - 0.5
    inner = np.s_[2:-2, 2:-2]
    assert np.abs(up[..., 0] - cx[None, :])[inner].max() < 1e-4 and np.abs(up[..., 1] - cy[:, None])[inner].max() < 1e-4


class _FixedDetector:
    def __init__(self, rows):
        self.rows = np.asarray(rows, dtype=np.float32)

    def detect_faces(self, image, **kw):
        self.seen = image.shape
        return self.rows


def test_restore_helper_host_half():
    """FaceRestoreHelper's host half (no device): the detector sees the INTER_AREA-reduced frame (resize=640), boxes / landmarks are
    scaled back, the reference's eye-distance rule (face_restoration_helper.py:223, indices as written) drops the tiny face,
    only_center_face / only_keep_largest select as the reference does, and the alignment fit maps every landmark set onto the template;
    the pixel steps refuse to run without a device instead of doing something else."""
    from codeformer_amd.facelib.utils.face_restoration_helper import _TEMPLATE_5, FaceRestoreHelper
    tpl = np.array(_TEMPLATE_5)

    def landmarks(cx, cy, size, ang):
        c, s = np.cos(ang), np.sin(ang)
        return ((tpl - 256) / 512 * size) @ np.array([[c, -s], [s, c]]).T + [cx, cy]

    scale = 640 / 720
    rows = []
    for cx, cy, size, ang in ((300, 260, 220, 0.15), (700, 420, 260, -0.2), (120, 600, 12, 0.0)):
        lm = landmarks(cx, cy, size, ang) * scale
        rows.append([lm[:, 0].min() - 20, lm[:, 1].min() - 30, lm[:, 0].max() + 20, lm[:, 1].max() + 20, 0.99] + lm.reshape(-1).tolist())
    det = _FixedDetector(rows)
    fh = FaceRestoreHelper(2, device='cpu', face_detector=det)
    frame = np.random.RandomState(0).randint(0, 255, (720, 960, 3)).astype(np.uint8)
    fh.read_image(frame)
    assert fh.get_face_landmarks_5(resize=640, eye_dist_threshold=5) == 2 and det.seen == (640, 853, 3)
    assert np.allclose(fh.all_landmarks_5[1], landmarks(700, 420, 260, -0.2), atol=1e-3) and len(fh.det_faces) == 2
    for lm, m in zip(fh.all_landmarks_5, fh.estimate_affines()):
        assert np.abs(lm @ m[:, :2].T + m[:, 2] - tpl).max() < 0.05          # similarity data: the fit reproduces the template
    with pytest.raises(RuntimeError):
        fh.align_warp_face()                                                 # crops are HIP kernels: no silent host substitute
    for kw, want in ((dict(only_center_face=True), (300, 260, 220, 0.15)), (dict(only_keep_largest=True), (700, 420, 260, -0.2))):
        fh.clean_all()
        fh.read_image(frame)
        assert fh.get_face_landmarks_5(resize=640, eye_dist_threshold=5, **kw) == 1
        assert np.allclose(fh.all_landmarks_5[0], landmarks(*want), atol=1e-3)
    # images whose short side is below 512 are enlarged first (read_image), gray inputs are flagged
    fh.clean_all()
    small = np.repeat(np.random.RandomState(1).randint(0, 255, (300, 400, 1)).astype(np.uint8), 3, axis=2)
    fh.read_image(small)
    assert fh.input_img.shape == (512, 683, 3) and fh.is_gray
    no_det = FaceRestoreHelper(1, device='cpu', face_detector=False)
    no_det.read_image(frame)
    with pytest.raises(RuntimeError):
        no_det.get_face_landmarks_5()
    with pytest.raises(NotImplementedError):
        FaceRestoreHelper(1, device='cpu', face_detector=False, pad_blur=True)
