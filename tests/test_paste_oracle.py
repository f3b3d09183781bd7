"""CPU: the numpy restatement of the OpenCV steps of the paste-back (oracle/paste_oracle.py) against independent implementations
(scipy.ndimage, exact real-valued bilinear interpolation) -- OpenCV itself is absent here, so this is what pins the oracle's
structure (window anchors, border rules, rounding); plus the frame sharding of codeformer_amd.video."""
import numpy as np
import scipy.ndimage as ndi

from oracle import paste_oracle as P


def _img(h, w, seed=0):
    rng = np.random.default_rng(seed)
    base = ndi.gaussian_filter(rng.normal(size=(h, w, 3)), (3, 3, 0)) * 400 + 128     # smooth structure + noise
    return np.clip(base + rng.normal(size=(h, w, 3)) * 8, 0, 255).astype(np.uint8)


def test_warp_identity_and_translation_are_exact():
    a = _img(40, 56)
    eye = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(P.warp_affine_u8(a, eye, (56, 40)), a)
    sh = np.array([[1.0, 0, 3], [0, 1.0, -2]])
    w = P.warp_affine_u8(a, sh, (56, 40), border_value=(9, 8, 7))
    assert np.array_equal(w[0:38, 3:], a[2:40, :53]) and w[39, 0].tolist() == [9, 8, 7]


def test_warp_matches_real_valued_bilinear_within_the_quantisation_bound():
    a = _img(64, 80, 1)
    m = P.similarity_from_points([[10, 10], [60, 12], [35, 50]], [[5, 8], [70, 20], [30, 70]])
    got = P.warp_affine_u8(a, m, (90, 100)).astype(np.float64)
    inv = P.invert_affine(m)
    ys, xs = np.mgrid[0:100, 0:90]
    sx, sy = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2], inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    ref = np.stack([ndi.map_coordinates(a[:, :, c].astype(np.float64), [sy, sx], order=1, mode='constant', cval=0.0) for c in range(3)], -1)
    inside = (sx > 1) & (sx < 78) & (sy > 1) & (sy < 62)
    # coordinates are truncated to 1/32 px after a +1/64 offset: at most 1/64 px off per axis -> |d| <= gradient/64*2 + rounding
    gy, gx = np.gradient(a.astype(np.float64), axis=(0, 1))
    bound = (np.abs(gx).max() + np.abs(gy).max()) / 32.0 + 1.0
    assert np.abs(got - ref)[inside].max() <= bound and np.abs(got - ref)[inside].mean() < 0.6
    f = P.warp_affine_f32(a[:, :, 0].astype(np.float32), m, (90, 100))
    assert np.abs(f - ref[:, :, 0])[inside].max() <= bound


def test_erode_matches_scipy_minimum_filter():
    rng = np.random.default_rng(2)
    x = rng.random((37, 41)).astype(np.float32)
    for k in (2, 3, 4, 7, 12):
        # OpenCV anchors an even window at k // 2: offsets [-(k//2), k-1-k//2]; scipy's origin shifts the centred window
        ref = ndi.minimum_filter(x, size=k, mode='constant', cval=np.inf, origin=0 if k % 2 else 0)
        got = P.erode(x, k)
        if k % 2:
            assert np.array_equal(got, ref), k
        else:   # scipy centres an even window on [-k/2, k/2-1], the same offsets as the anchor k//2
            assert np.array_equal(got, ref), k
    assert np.array_equal(P.erode(x, 1), x) and np.array_equal(P.erode(x, 0), P.erode(x, 3))


def test_gaussian_blur_matches_scipy_with_mirror_border():
    rng = np.random.default_rng(3)
    x = rng.random((45, 33)).astype(np.float32)
    for ksize in (3, 9, 31):
        k = P.gaussian_kernel(ksize).astype(np.float64)
        assert abs(k.sum() - 1) < 1e-6 and np.allclose(k, k[::-1])
        ref = ndi.correlate1d(ndi.correlate1d(x.astype(np.float64), k, axis=1, mode='mirror'), k, axis=0, mode='mirror')   # reflect-101
        assert np.abs(P.gaussian_blur(x, ksize) - ref).max() < 1e-5, ksize
    assert abs(0.3 * ((31 - 1) * 0.5 - 1) + 0.8 - 5.0) < 1e-12        # the sigma rule at ksize 31
    assert np.allclose(P.gaussian_kernel(101, 11.0)[50], 1.0 / (11.0 * np.sqrt(2 * np.pi)), rtol=1e-3)


def test_resize_equals_the_product_host_restatement():
    from codeformer_amd.utils.img_util import resize_bilinear
    a = _img(30, 44, 4)
    for size in ((88, 60), (22, 15), (44, 30), (51, 37)):
        assert np.array_equal(P.resize_linear_u8(a, size), resize_bilinear(a, size)), size


def test_paste_leaves_the_frame_alone_outside_the_face_and_takes_the_face_inside():
    frame = _img(240, 320, 5)
    face = _img(512, 512, 6)
    aff = P.similarity_from_points([[140, 100], [180, 100], [160, 140]], [[192, 240], [320, 240], [256, 368]])     # ~3.2x zoom: a 160 px face
    out = P.paste_faces(frame, [face], [aff], upscale=1)
    inv = P.invert_affine(aff)
    centre = (inv @ np.array([256, 256, 1.0])).round().astype(int)
    assert np.array_equal(out[:5], frame[:5]) and np.array_equal(out[:, :5], frame[:, :5])            # far from the face: untouched
    crop = P.align_warp_face(out, aff)
    assert np.abs(crop[200:312, 200:312].astype(int) - face[200:312, 200:312].astype(int)).mean() < 12   # the face landed where it came from
    assert out.dtype == np.uint8 and 0 <= centre[0] < 320


def test_frame_shard_partitions_the_clip():
    from codeformer_amd.video import frame_shard
    for n in (0, 1, 7, 300):
        for world in (1, 2, 8):
            seen = [i for r in range(world) for i in frame_shard(n, r, world)]
            assert seen == list(range(n))
    assert list(frame_shard(300, 7, 8)) == list(range(263, 300))
