"""-m gpu: alignment warp, paste-back and the video fan-out on the device (cf_paste.hip through the C ABI) against the numpy
restatement of the reference's OpenCV steps (oracle/paste_oracle.py; face_restoration_helper.py:320-499).  Integer / uint8 results
must be bit-exact; float32 masks are produced with the same separately-rounded operations in the same order, so they are too."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _img(h, w, seed=0):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(seed)
    base = ndi.gaussian_filter(rng.normal(size=(h, w, 3)), (3, 3, 0)) * 400 + 128
    return np.clip(base + rng.normal(size=(h, w, 3)) * 8, 0, 255).astype(np.uint8)


def _affine(P, cx, cy, size, angle=0.0):
    """frame -> 512-face similarity for a face of `size` frame pixels centred at (cx, cy), rotated by `angle`."""
    c, s = np.cos(angle), np.sin(angle)
    pts = np.array([[-0.25, -0.1], [0.25, -0.1], [0.0, 0.25]]) * size
    src = pts @ np.array([[c, -s], [s, c]]).T + [cx, cy]
    dst = np.array([[-0.25, -0.1], [0.25, -0.1], [0.0, 0.25]]) * 512 + 256
    return P.similarity_from_points(src, dst)


@pytest.fixture(scope='module')
def env():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from codeformer_amd import lib, ops
    from oracle import paste_oracle as P
    lib.load()
    return torch, ops, P


def test_primitives_are_bit_exact(env):
    torch, ops, P = env
    from codeformer_amd.facelib.paste import gaussian_taps, invert_affine
    a = _img(90, 130, 1)
    mats = [_affine(P, 60, 40, 70, 0.3), np.array([[0.9, 0.2, -7.5], [-0.15, 1.1, 4.25]]), np.array([[1.0, 0, 0], [0, 1.0, 0]])]
    inv = torch.from_numpy(np.stack([invert_affine(m) for m in mats]).reshape(3, 6)).cuda()
    dst = torch.zeros(3, 100, 120, 3, dtype=torch.uint8, device='cuda')
    ops.warp_affine_u8(torch.from_numpy(a).cuda(), inv, dst, border=(135, 133, 132))
    for i, m in enumerate(mats):
        assert np.array_equal(dst[i].cpu().numpy(), P.warp_affine_u8(a, m, (120, 100), border_value=(135, 133, 132))), i
    # region form: only the window is written
    d2 = torch.full((1, 100, 120, 3), 7, dtype=torch.uint8, device='cuda')
    ops.warp_affine_u8(torch.from_numpy(a).cuda(), inv[:1], d2, region=(10, 20, 50, 30))
    ref = np.full((100, 120, 3), 7, np.uint8)
    ref[20:50, 10:60] = P.warp_affine_u8(a, mats[0], (120, 100))[20:50, 10:60]
    assert np.array_equal(d2[0].cpu().numpy(), ref)
    f = np.random.default_rng(2).random((64, 48)).astype(np.float32)
    got = ops.warp_affine_f32(torch.from_numpy(f).cuda(), invert_affine(mats[1]), (5, 3, 80, 60)).cpu().numpy()
    assert np.array_equal(got, P.warp_affine_f32(f, mats[1], (100, 70))[3:63, 5:85])
    x = np.random.default_rng(3).random((75, 61)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    for k in (0, 1, 2, 3, 4, 12, 31):
        assert np.array_equal(ops.erode(xd, k).cpu().numpy(), P.erode(x, k)), k
    for ksize in (1, 3, 7, 21, 61):
        taps = torch.from_numpy(gaussian_taps(ksize)).cuda()
        assert np.array_equal(gaussian_taps(ksize), P.gaussian_kernel(ksize))
        assert np.array_equal(ops.gaussian_blur(xd, taps).cpu().numpy(), P.gaussian_blur(x, ksize)), ksize
    # a window of a larger frame whose outside is zero == the same window of the full-frame blur
    full = np.zeros((120, 100), np.float32)
    full[30:105, 20:81] = x
    taps = torch.from_numpy(gaussian_taps(21)).cuda()
    assert np.array_equal(ops.gaussian_blur(xd, taps, (20, 30), (120, 100)).cpu().numpy(), P.gaussian_blur(full, 21)[30:105, 20:81])
    for size in ((260, 180), (65, 45), (130, 90)):
        assert np.array_equal(ops.resize_linear_u8(torch.from_numpy(a).cuda(), size[1], size[0]).cpu().numpy(),
                              P.resize_linear_u8(a, size).astype(np.float32)), size
    part = torch.empty(64, dtype=torch.float64, device='cuda')
    ops.sum_partials(xd, part)
    assert abs(float(part.sum()) - float(x.astype(np.float64).sum())) < 1e-9
    assert np.array_equal(ops.f32_to_u8_trunc(torch.tensor([0.0, 0.99, 1.0, 254.7, 255.0], device='cuda')).cpu().numpy(), [0, 0, 1, 254, 255])


@pytest.mark.parametrize('upscale', [1, 2])
def test_align_and_paste_match_the_oracle(env, upscale):
    torch, ops, P = env
    from codeformer_amd.facelib.paste import DeviceFaceHelper
    frame = _img(270, 480, 10 + upscale)
    affs = [_affine(P, 150, 120, 130, 0.2), _affine(P, 260, 150, 150, -0.15), _affine(P, 470, 20, 90, 0.0)]   # overlapping pair + one cut by the border
    h = DeviceFaceHelper(upscale_factor=upscale, device='cuda')
    h.read_image(frame)
    crops = h.align_warp_face(affs)
    for i, a in enumerate(affs):
        assert np.array_equal(crops[i].cpu().numpy(), P.align_warp_face(frame, a)), i
    restored = [_img(512, 512, 20 + i) for i in range(3)]
    h.add_restored_faces(torch.from_numpy(np.stack(restored)).cuda())
    got = h.paste_faces_to_input_image()
    want = P.paste_faces(frame, restored, affs, upscale=upscale)
    assert got.shape == want.shape == (270 * upscale, 480 * upscale, 3)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() == 0, (int(d.max()), float((d > 0).mean()))
    assert (got != P.resize_linear_u8(frame, (480 * upscale, 270 * upscale)) if upscale > 1 else got != frame).any()


def test_parse_mask_branch(env):
    torch, ops, P = env
    from codeformer_amd.facelib.paste import DeviceFaceHelper
    rng = np.random.default_rng(5)
    labels = rng.integers(0, 19, (2, 512, 512))
    labels[:, 100:400, 120:390] = 1                       # a face-like blob of "skin"

    class FakeParse:
        def parse_labels(self, x):
            return torch.from_numpy(labels).cuda()

    h = DeviceFaceHelper(upscale_factor=1, device='cuda', use_parse=True, face_parse=FakeParse())
    frame = _img(200, 300, 30)
    affs = [_affine(P, 100, 90, 120, 0.1), _affine(P, 210, 110, 100, -0.2)]
    restored = [_img(512, 512, 40), _img(512, 512, 41)]
    h.read_image(frame)
    h.align_warp_face(affs)
    h.add_restored_faces(torch.from_numpy(np.stack(restored)).cuda())
    soft = h.parse_soft_masks(h.restored_faces).cpu().numpy()
    ref_soft = [P.parse_soft_mask(labels[i]) for i in range(2)]
    for i in range(2):
        assert np.abs(soft[i] - ref_soft[i]).max() <= 2e-7, i      # 1/255 is applied after the blurs here, inside them never
    got = h.paste_faces_to_input_image()
    want = P.paste_faces(frame, restored, affs, upscale=1, parse_masks=[soft[0], soft[1]])
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() == 0
    plain = P.paste_faces(frame, restored, affs, upscale=1)
    assert (want != plain).any()


def test_video_fanout_batches_faces_across_frames(env):
    torch, ops, P = env
    from codeformer_amd.video import VideoRestorer
    rng = np.random.default_rng(7)
    frames = [_img(180, 320, 50 + i) for i in range(9)]
    affs = []
    for i in range(9):
        k = [3, 0, 5, 2, 4, 1, 6, 2, 3][i]
        affs.append(np.stack([_affine(P, rng.uniform(60, 260), rng.uniform(50, 130), rng.uniform(50, 110), rng.uniform(-0.3, 0.3))
                              for _ in range(k)]) if k else np.zeros((0, 2, 3)))
    sizes = []

    def net(x, w=0.5, adain=True):                       # a recognisable per-face function: the negative image
        sizes.append(x.shape[0])
        return (-x,)

    vr = VideoRestorer(net, 'cuda', upscale=1, batch_size=8)
    out = vr.restore(frames, affs, w=0.5)
    assert sizes == [8, 8, 8, 2] and vr.stats == {'frames': 9, 'faces': 26, 'forward_calls': 4}     # full batches across frame borders
    for i in range(9):
        crops = [P.align_warp_face(frames[i], a) for a in affs[i]]
        want = P.paste_faces(frames[i], [255 - c for c in crops], list(affs[i]), upscale=1)
        assert np.array_equal(out[i], want), i


def test_entrypoint_whole_images_with_host_affines(env, tmp_path):
    """inference_codeformer.py without --has_aligned: the host detector's alignment matrices come in through --affine_npz, crop
    warp / restoration / paste-back run on the GPU, final_results/<name>.png is written at the upscaled size."""
    import subprocess
    import sys
    from PIL import Image
    torch, ops, P = env
    src = tmp_path / 'whole_imgs'
    os.makedirs(src)
    table = {}
    for i in range(2):
        Image.fromarray(_img(520, 640, 70 + i)[:, :, ::-1]).save(src / f'im{i}.png')
        table[f'im{i}'] = np.stack([_affine(P, 220 + 60 * i, 210, 180, 0.1 * i)])
    np.savez(tmp_path / 'aff.npz', **table)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '-i', str(src), '-o', str(tmp_path / 'o'), '-s', '2',
                        '--device', 'cuda', '--random_init_seed', '0', '--affine_npz', str(tmp_path / 'aff.npz')],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(tmp_path / 'o' / 'final_results')) == ['im0.png', 'im1.png']
    out = np.asarray(Image.open(tmp_path / 'o' / 'final_results' / 'im0.png'))
    assert out.shape == (1040, 1280, 3) and '2 faces of 2 images in 1 forward calls' in r.stdout
    assert sorted(os.listdir(tmp_path / 'o' / 'cropped_faces')) == ['im0_00.png', 'im1_00.png']
    assert sorted(os.listdir(tmp_path / 'o' / 'restored_faces')) == ['im0_00.png', 'im1_00.png']
    crop = np.asarray(Image.open(tmp_path / 'o' / 'cropped_faces' / 'im0_00.png'))[:, :, ::-1]
    frame = np.asarray(Image.open(src / 'im0.png'))[:, :, ::-1]
    assert np.array_equal(crop, P.align_warp_face(np.ascontiguousarray(frame), table['im0'][0]))


def test_area_reduction_kernel_is_bit_exact(env):
    """cf_resize_area_u8 (cv2.resize INTER_AREA for the detector's input) against its numpy restatement: fractional ratios (float32
    accumulation in OpenCV's order), integer ratios, the 2 x 2 rule, identity; enlarging is refused."""
    torch, ops, P = env
    from basicsr.utils.img_util import resize_area
    for (h, w, dh, dw, seed) in ((1080, 1920, 640, 1137, 1), (720, 960, 640, 853, 2), (531, 777, 200, 301, 3), (96, 120, 32, 40, 4),
                                 (96, 120, 48, 60, 5), (90, 120, 30, 40, 6), (64, 64, 64, 64, 7), (513, 640, 512, 639, 8)):
        img = _img(h, w, seed)
        got = ops.resize_area_u8(torch.from_numpy(img).cuda(), dh, dw).cpu().numpy()
        want = resize_area(img, (dw, dh))
        assert got.shape == want.shape == (dh, dw, 3) and np.array_equal(got, want), (h, w, dh, dw, int(np.abs(got.astype(int) - want).max()))
    with pytest.raises(RuntimeError):
        ops.resize_area_u8(torch.zeros(10, 10, 3, dtype=torch.uint8, device='cuda'), 11, 10)


class _FixedDetector:
    """Stands in for RetinaFace.detect_faces: returns prepared (k, 15) rows for whatever image it is given."""

    def __init__(self, rows):
        self.rows = np.asarray(rows, dtype=np.float32)

    def detect_faces(self, image, **kw):
        self.seen = image.shape
        return self.rows


def test_face_restore_helper_host_detection_device_pixels(env):
    """FaceRestoreHelper end to end with a prepared detector output: landmarks -> LMedS similarity (host) -> crops (device, one
    launch) -> paste-back (device) == the numpy restatement driven by the same matrices; detector input is the INTER_AREA-reduced frame
    and its boxes are scaled back; the too-small-eye-distance rule and only_center_face apply."""
    torch, ops, P = env
    from codeformer_amd.facelib.align import estimate_affine_partial_2d
    from codeformer_amd.facelib.utils.face_restoration_helper import _TEMPLATE_5, FaceRestoreHelper
    frame = _img(720, 960, 5)
    tpl = np.array(_TEMPLATE_5)

    def landmarks(cx, cy, size, ang):
        c, s = np.cos(ang), np.sin(ang)
        return ((tpl - 256) / 512 * size) @ np.array([[c, -s], [s, c]]).T + [cx, cy]

    scale = 640 / 720
    rows = []
    for cx, cy, size, ang in ((300, 260, 220, 0.15), (700, 420, 260, -0.2), (120, 600, 12, 0.0)):      # the third one is tiny
        lm = landmarks(cx, cy, size, ang) * scale
        box = [lm[:, 0].min() - 20, lm[:, 1].min() - 30, lm[:, 0].max() + 20, lm[:, 1].max() + 20, 0.99]
        rows.append(box + lm.reshape(-1).tolist())
    det = _FixedDetector(rows)
    fh = FaceRestoreHelper(2, face_size=512, det_model='retinaface_resnet50', use_parse=False, device='cuda', face_detector=det)
    fh.read_image(frame)
    n = fh.get_face_landmarks_5(resize=640, eye_dist_threshold=5)
    assert det.seen == (640, 853, 3) and n == 2                       # the 12-pixel face is dropped by the eye-distance rule
    assert np.allclose(fh.all_landmarks_5[1], landmarks(700, 420, 260, -0.2), atol=1e-3)
    fh.align_warp_face()
    crops = fh.cropped_faces
    assert len(crops) == 2 and crops[0].shape == (512, 512, 3)
    for k in range(2):
        m = estimate_affine_partial_2d(fh.all_landmarks_5[k], tpl)[0]
        assert np.array_equal(fh.affine_matrices[k], m)
        assert np.array_equal(crops[k], P.align_warp_face(frame, m))
    restored = 255 - fh.cropped_faces_device
    fh.add_restored_faces(restored)
    fh.get_inverse_affine()
    out = fh.paste_faces_to_input_image()
    want = P.paste_faces(frame, [255 - c for c in crops], list(fh.affine_matrices), upscale=2)
    assert out.shape == (1440, 1920, 3) and np.array_equal(out, want)
    # only_center_face keeps the detection nearest to the image centre
    fh.clean_all()
    fh.read_image(frame)
    assert fh.get_face_landmarks_5(only_center_face=True, resize=640, eye_dist_threshold=5) == 1
    assert np.allclose(fh.all_landmarks_5[0], landmarks(300, 260, 220, 0.15), atol=1e-3)      # box centre 203 px from (480, 360); the other 231
    # a detector that lives on the device is handed the device frame, reduced by the INTER_AREA kernel: nothing but boxes crosses PCIe
    from basicsr.utils.img_util import resize_area
    det.device = torch.device('cuda')
    fh.clean_all()
    fh.read_image(frame)
    det.seen_input = None
    orig = det.detect_faces
    det.detect_faces = lambda image, **kw: (setattr(det, 'seen_input', image), orig(image, **kw))[1]
    assert fh.get_face_landmarks_5(resize=640, eye_dist_threshold=5) == 2
    assert torch.is_tensor(det.seen_input) and det.seen_input.is_cuda and det.seen_input.dtype == torch.uint8
    assert np.array_equal(det.seen_input.cpu().numpy(), resize_area(frame, (853, 640)))
    det.detect_faces = orig
    del det.device
    fh.clean_all()
    fh.read_image(frame)
    assert fh.get_face_landmarks_5(only_center_face=True, resize=640, eye_dist_threshold=5) == 1
    # per-face host path of the reference's loop (add_restored_face with numpy arrays)
    fh.align_warp_face()
    fh.add_restored_face(255 - fh.cropped_faces[0], fh.cropped_faces[0])
    out1 = fh.paste_faces_to_input_image()
    want1 = P.paste_faces(frame, [255 - fh.cropped_faces[0]], list(fh.affine_matrices), upscale=2)
    assert np.array_equal(out1, want1)


def test_entrypoint_whole_images_with_host_detector(env, tmp_path):
    """The reference's default invocation (no --has_aligned, no matrices file): RetinaFace runs on the host.  With seeded random
    weights the detections are meaningless, but the flow -- read, detect at 640, fit, warp, restore, paste, result tree -- is the real one."""
    import subprocess
    import sys
    from PIL import Image
    src = tmp_path / 'whole_imgs'
    os.makedirs(src)
    Image.fromarray(_img(300, 400, 90)[:, :, ::-1]).save(src / 'a.png')         # short side below 512: enlarged to 512 x 683 first
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '-i', str(src), '-o', str(tmp_path / 'o'), '-s', '1',
                        '--device', 'cuda', '--random_init_seed', '0', '--detection_model', 'retinaface_mobile0.25'],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'detect ' in r.stdout and os.listdir(tmp_path / 'o' / 'final_results') == ['a.png']
    assert np.asarray(Image.open(tmp_path / 'o' / 'final_results' / 'a.png')).shape == (512, 683, 3)
