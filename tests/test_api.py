"""Plugin boundary of the path: registry semantics, module API, host path == reference golden (CPU)."""
import os

import numpy as np
import pytest
import torch

from oracle.synth import seeded_input


def test_registry_semantics():
    from basicsr.utils.registry import ARCH_REGISTRY, Registry
    import basicsr.archs  # noqa: F401
    assert 'CodeFormer' in ARCH_REGISTRY and 'VQAutoEncoder' in ARCH_REGISTRY
    with pytest.raises(KeyError, match="No object named 'Nope' found in 'arch' registry!"):
        ARCH_REGISTRY.get('Nope')
    r = Registry('t')

    @r.register()
    class A:
        pass

    assert r.get('A') is A and list(r.keys()) == ['A'] and dict(iter(r)) == {'A': A}
    with pytest.raises(AssertionError):
        r.register(A)


def test_drop_in_import_paths():
    from basicsr.archs.codeformer_arch import CodeFormer, Fuse_sft_block, TransformerSALayer  # noqa: F401
    from basicsr.archs.vqgan_arch import (AttnBlock, Downsample, Encoder, Generator, ResBlock, Upsample,  # noqa: F401
                                          VectorQuantizer, VQAutoEncoder)
    from basicsr.utils import get_root_logger, img2tensor, imwrite, tensor2img  # noqa: F401
    from basicsr.utils.misc import get_device, gpu_is_available
    assert str(get_device()) in ('cpu', 'cuda')
    assert gpu_is_available() in (True, False)
    with pytest.raises(TypeError):
        get_device('0')


def test_constructor_variants_and_key_sets(seed0_net):
    from basicsr.utils.registry import ARCH_REGISTRY
    keys = set(seed0_net.state_dict())
    assert len(keys) == 515 and 'fuse_convs_dict.256.shift.2.bias' in keys and 'idx_pred_layer.1.weight' in keys
    assert sum(p.numel() for p in seed0_net.parameters()) == 94112707
    assert not any(p.requires_grad for p in seed0_net.generator.parameters())  # fix_modules
    inp = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=512, n_head=8, n_layers=9, connect_list=['32', '64', '128'])
    assert inp.quantize.embedding.weight.shape == (512, 256) and '256' not in inp.fuse_convs_dict
    assert inp.idx_pred_layer[1].weight.shape == (512, 512)


def test_host_path_matches_reference_golden(seed0_net, golden_dir):
    """device='cpu' plumbing path (BASELINE config 1) reproduces the reference bit-for-bit on this torch build."""
    g = np.load(os.path.join(golden_dir, 'restoration_seed0_face0.npz'))
    with torch.no_grad():
        out, logits, lq = seed0_net(seeded_input(1), w=0.5, adain=True)
    assert float((out - torch.from_numpy(g['out'])).abs().max()) <= 1e-4
    assert np.array_equal(logits.argmax(-1).numpy(), g['idx'])
    assert out.shape == (1, 3, 512, 512) and logits.shape == (1, 256, 1024) and lq.shape == (1, 256, 16, 16)
    with torch.no_grad():
        lg2, lq2 = seed0_net(seeded_input(1), w=0.5, code_only=True)
    assert torch.equal(lg2, logits) and torch.equal(lq2, lq)


def test_img_util_round_trip_semantics():
    from basicsr.utils import img2tensor, tensor2img
    from basicsr.utils.img_util import normalize_
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (8, 6, 3), dtype=np.uint8)
    t = img2tensor(img / 255., bgr2rgb=True, float32=True)
    assert t.dtype == torch.float32 and t.shape == (3, 8, 6)
    assert torch.equal(t[0], torch.from_numpy((img[:, :, 2] / 255.).astype(np.float32)))   # BGR -> RGB
    normalize_(t, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    assert float(t.min()) >= -1 and float(t.max()) <= 1
    back = tensor2img(t.unsqueeze(0), rgb2bgr=True, min_max=(-1, 1))
    assert back.dtype == np.uint8 and np.array_equal(back, img)
    # half-to-even rounding + clamp
    z = torch.tensor([[[0.5 / 255 * 2 - 1, 1.5 / 255 * 2 - 1, 7.0, -7.0]]]).repeat(3, 1, 1)
    assert tensor2img(z, min_max=(-1, 1))[0, :, 0].tolist() in ([0, 2, 255, 0], [1, 2, 255, 0], [0, 1, 255, 0], [1, 1, 255, 0])
    assert len(img2tensor([img / 255., img / 255.])) == 2
    with pytest.raises(TypeError):
        tensor2img(np.zeros((3, 4, 4)))


def test_face_misc_helpers():
    from codeformer_amd.utils.face_misc import AlignedFaceHelper, adain_npy, bgr2gray, is_gray
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    assert is_gray(np.stack([g, g, g], -1)) and not is_gray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8))
    col = rng.integers(0, 256, (16, 16, 3)).astype(np.float64)
    out = adain_npy(bgr2gray(col), col)
    assert np.allclose(out.reshape(-1, 3).mean(0), col.reshape(-1, 3).mean(0))
    h = AlignedFaceHelper()
    h.is_gray = True
    h.add_restored_face(col, col)
    assert h.restored_faces[0].shape == (16, 16, 3)
    h.clean_all()
    assert h.restored_faces == [] and h.cropped_faces == []


def test_bundled_ops_shim_cpu():
    """`basicsr.ops` names of the reference import here; CPU upfirdn2d follows the oracle, grads are refused."""
    import pytest
    import torch
    from basicsr.ops.fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
    from basicsr.ops.upfirdn2d import upfirdn2d
    from oracle import codeformer_oracle as O
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 9, 11, generator=g)
    k = torch.randn(4, 4, generator=g)
    for up, down, pad in [(1, 1, (0, 0)), (2, 1, (2, 1)), (1, 2, (1, 1)), (2, 3, (3, 2)), (1, 1, (-1, 2))]:
        got = upfirdn2d(x, k, up=up, down=down, pad=pad)
        ref = O.upfirdn2d(x, k, up=up, down=down, pad=pad)
        assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-5), (up, down, pad)
    m = FusedLeakyReLU(3)
    assert list(m.state_dict()) == ['bias'] and m.negative_slope == 0.2 and abs(m.scale - 2 ** 0.5) < 1e-12
    with pytest.raises(RuntimeError):
        upfirdn2d(x.requires_grad_(True), k)
    with pytest.raises(RuntimeError):
        fused_leaky_relu(x.detach(), m.bias.detach())   # no CPU implementation, as in the reference


def test_winograd_eligibility_rule():
    """Which 3x3 stride-1 layers take the Winograd kernel is a pure function of the layer shape (never of the batch): whole 8x16
    output patches, cin % 16 == 0, cout % 64 == 0.  Every such conv of CodeFormer's encoder / generator / fusion blocks qualifies;
    the 3-channel stem / head and odd sizes do not."""
    from codeformer_amd import ops
    for cin, cout, h in ((64, 64, 512), (128, 128, 256), (128, 128, 128), (256, 256, 64), (256, 256, 32), (512, 512, 16),
                         (256, 512, 16), (512, 256, 32), (128, 64, 512), (64, 128, 256)):
        assert ops.winograd_ok(cin, cout, h, h), (cin, cout, h)
    assert not ops.winograd_ok(3, 64, 512, 512) and not ops.winograd_ok(64, 3, 512, 512)
    assert not ops.winograd_ok(64, 64, 20, 32) and not ops.winograd_ok(64, 64, 16, 24) and not ops.winograd_ok(64, 96, 16, 16)
    assert ops.WINOGRAD == 3


def test_f43_eligibility_rule(monkeypatch):
    """Which generator / fusion layers a SPLIT_F43 request puts on the F(4x4,3x3) kernel (ops.f43_ok): a pure function of the layer
    shape -- whole 16x16 patches, cin % 16 == 0 and <= 256 (<= 512 in the 16-wave form on 32-channel slabs, round 6), cout % 64 == 0 -- narrowed by CODEFORMER_HIP_F43 ('auto': 64 output channels,
    or a multiple of 128 from 128x128 pixels up (fp32 operands: 64x64); 'c64'; 'all'; '0'); conv_code falls back to the F(2x2,3x3) split-half kernel elsewhere
    and never returns WF43 for a plain SPLIT request (the encoder)."""
    from codeformer_amd import ops
    monkeypatch.setattr(ops, 'F43_LAYERS', 'auto')
    for cin, cout, h, want in ((64, 64, 512, True), (128, 64, 512, True), (128, 128, 256, True), (256, 128, 256, True), (128, 128, 128, True),
                               (256, 256, 128, True), (256, 256, 32, False), (512, 256, 128, True), (512, 64, 128, False), (768, 256, 128, False), (64, 128, 24, False), (128, 192, 256, False),
                               (48, 64, 32, True), (40, 64, 32, False)):
        assert ops.f43_ok(cin, cout, h, h) == want, (cin, cout, h)
        assert (ops.conv_code(ops.SPLIT_F43, cin, cout, h, h) == ops.WF43) == want
        assert ops.conv_code(ops.SPLIT, cin, cout, h, h) != ops.WF43
        # the exact-fp32 request of precision='fp32' (generator / fusion only): the same rule, fp32 operands; a plain WINOGRAD request never gets it
        assert (ops.conv_code(ops.WINOGRAD_F43, cin, cout, h, h) == ops.WF43F) == want
        assert ops.conv_code(ops.WINOGRAD, cin, cout, h, h) != ops.WF43F
        assert ops.conv_code(ops.WINOGRAD_F43, cin, cout, h, h, up2x=True) not in (ops.WF43, ops.WF43F)
    # 64x64 images: the 16-wave form only with fp32 operands (split halves: from 128x128 up -- one-face latency)
    assert not ops.f43_ok(256, 256, 64, 64) and ops.f43_ok(256, 256, 64, 64, fp32=True) and ops.f43_ok(256, 64, 64, 64)
    assert ops.conv_code(ops.SPLIT_F43, 256, 256, 64, 64) == ops.WSPLIT and ops.conv_code(ops.WINOGRAD_F43, 256, 256, 64, 64) == ops.WF43F
    # a concat boundary must not cut a slab of the form that runs: 32 channels where cout % 128 == 0 and cin % 32 == 0, else 16
    assert ops.conv_code(ops.SPLIT_F43, 128, 128, 256, 256, c_split=64) == ops.WF43 and ops.conv_code(ops.SPLIT_F43, 128, 128, 256, 256, c_split=48) != ops.WF43
    assert ops.conv_code(ops.SPLIT_F43, 128, 64, 256, 256, c_split=48) == ops.WF43 and ops.conv_code(ops.WINOGRAD_F43, 128, 64, 256, 256, c_split=40) == ops.WINOGRAD
    assert ops.exact_code(ops.WINOGRAD_F43) == ops.WINOGRAD_F43 and not ops.needs_act_scale(ops.PackedWeight(None, None, 64, 64, 9, 64, 64, wino=2))
    monkeypatch.setattr(ops, 'F43_LAYERS', 'c64')
    assert ops.f43_ok(64, 64, 512, 512) and not ops.f43_ok(128, 128, 256, 256)
    monkeypatch.setattr(ops, 'F43_LAYERS', 'all')
    assert ops.f43_ok(256, 256, 32, 32) and ops.f43_ok(512, 256, 64, 64) and not ops.f43_ok(512, 64, 64, 64) and not ops.f43_ok(768, 256, 64, 64)
    monkeypatch.setattr(ops, 'F43_LAYERS', '0')
    assert not ops.f43_ok(64, 64, 512, 512)
    assert ops.conv_code(ops.SPLIT_F43, 64, 64, 512, 512) == ops.WSPLIT
