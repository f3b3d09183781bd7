"""The CPU oracle against fixtures produced by the REFERENCE's own arch files (oracle/make_golden.py).

These run on CPU (-m "not gpu") and are what pins the oracle: the reference ships no tests of its own.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import codeformer_oracle as O
from oracle import ref_loader
from oracle.synth import seeded_input, seeded_randn, synth_state_dict


def _sd(net):
    return {k: v.detach() for k, v in net.state_dict().items()}


def test_seed0_weights_match_reference_digest(seed0_net, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, 'state_dict_seed0_digest.json')))
    sd = _sd(seed0_net)
    assert set(sd) == set(ref)
    for k, v in sd.items():
        assert hashlib.sha256(v.contiguous().numpy().tobytes()).hexdigest()[:16] == ref[k], k
    # known-answer values recorded from the reference
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - 1218901.639128) < 1e-3
    assert torch.allclose(sd['encoder.blocks.0.weight'][0, 0, 0], torch.tensor([-0.00144084, 0.10323862, -0.15839511]), atol=1e-7)


def test_oracle_full_forward_matches_reference_golden(seed0_net, golden_dir):
    g = np.load(os.path.join(golden_dir, 'restoration_seed0_face0.npz'))
    x = seeded_input(1)
    assert abs(float(x[0, 0, 0, 0]) + 0.94204152) < 1e-6
    out, logits, lq, idx = O.codeformer_forward(x, _sd(seed0_net), w=0.5, adain_flag=True, return_idx=True)
    assert float((lq - torch.from_numpy(g['lq_feat'])).abs().max()) <= 1e-5
    assert float((logits - torch.from_numpy(g['logits'])).abs().max()) <= 2e-5
    assert np.array_equal(idx.numpy(), g['idx'])
    assert hashlib.sha256(idx.numpy().astype('<i8').tobytes()).hexdigest()[:16] == '7d2fd85619ab8528'
    assert float((out - torch.from_numpy(g['out'])).abs().max()) <= 1e-4
    assert abs(float(out.mean()) - 0.07784) < 1e-4 and abs(float(out.abs().mean()) - 0.37604) < 1e-4


def test_oracle_blocks_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'blocks_seed7.npz'))
    shapes = json.load(open(os.path.join(golden_dir, 'blocks_seed7_shapes.json')))
    sd = synth_state_dict(shapes, 7)
    xr, xa = seeded_randn((1, 64, 32, 32), 71), seeded_randn((1, 512, 16, 16), 72)
    xt, pos = seeded_randn((256, 2, 512), 73), seeded_randn((256, 1, 512), 74).repeat(1, 2, 1)
    xe, xd, xs = seeded_randn((1, 128, 32, 32), 75), seeded_randn((1, 128, 32, 32), 76), seeded_randn((1, 64, 32, 32), 77)
    got = {
        'res': O.res_block(xr, sd, 'res'), 'attn': O.attn_block(xa, sd, 'attn'),
        'tl': O.transformer_layer(xt, pos, sd, 'tl', 8), 'fuse': O.fuse_sft(xe, xd, 0.7, sd, 'fuse'),
        'down': O.downsample(xs, sd, 'down'), 'up': O.upsample(xs, sd, 'up'), 'adain': O.adain(xe, xd),
    }
    for k, v in got.items():
        assert float((v - torch.from_numpy(g['out_' + k])).abs().max()) <= 2e-5, k


def test_oracle_vq_nearest_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'vq_seed11.npz'))
    zq, idx, _ = O.vq_nearest(torch.from_numpy(g['z']), torch.from_numpy(g['codebook']))
    assert np.array_equal(idx.numpy(), g['idx'])
    # the reference returns z + (z_q - z) (straight-through form): equal to the gathered rows up to one rounding
    assert float((zq - torch.from_numpy(g['zq'])).abs().max()) <= 1e-9


def test_oracle_recorded_agreement_with_reference(golden_dir):
    """The generator script recorded oracle-vs-reference deltas when the goldens were made."""
    rep = json.load(open(os.path.join(golden_dir, 'oracle_vs_reference.json')))
    for tag in ('w0.5', 'w0.0', 'w1.0'):
        assert rep[tag]['oracle_vs_ref_out'] <= 1e-5 and rep[tag]['oracle_idx_equal']
        assert rep[tag]['oracle_vs_ref_logits'] <= 1e-5
    assert rep['inpaint']['idx_equal'] and rep['inpaint']['oracle_vs_ref_out'] <= 1e-5
    assert all(v <= 1e-6 for v in rep['blocks'].values())


def test_tensor2img_u8_rounding():
    t = torch.tensor([[[-1.0, 1.0, 0.0, 1.0 / 255.0 - 1.0 + 1e-9, 2.0, -3.0]]]).repeat(3, 1, 1)
    img = O.tensor2img_u8(t)
    assert img.dtype == np.uint8 and img.shape == (1, 6, 3)
    assert img[0, :, 0].tolist() == [0, 255, 128, 0, 255, 0]   # 127.5 -> 128 (half-to-even), clamp


def test_oracle_bundled_ops_restatements():
    x = seeded_randn((2, 3, 8, 8), 5)
    y = O.fused_bias_act(x, torch.tensor([0.1, -0.2, 0.3]))
    ref = torch.nn.functional.leaky_relu(x + torch.tensor([0.1, -0.2, 0.3]).view(1, 3, 1, 1), 0.2) * 2 ** 0.5
    assert torch.equal(y, ref)
    k = torch.tensor([[1., 3., 3., 1.]]).t() @ torch.tensor([[1., 3., 3., 1.]]) / 64
    up = O.upfirdn2d(x, k * 4, up=2, pad=(2, 1))
    assert up.shape == (2, 3, 16, 16)
    dn = O.upfirdn2d(x, k, down=2, pad=(1, 1))
    assert dn.shape == (2, 3, 4, 4)


@pytest.mark.skipif(not ref_loader.available(), reason='reference tree not mounted (GPU box)')
def test_oracle_fused_bias_act_modes():
    """The act*10+grad switch (fused_bias_act_kernel.cu:36-46): mode 30 is the forward formula, mode 31 its derivative w.r.t. the
    pre-activation gated by the forward output, 12/32 are zero, 10/11 linear."""
    x, b = torch.randn(2, 3, 4, 5, generator=torch.Generator().manual_seed(5)), torch.tensor([0.1, -0.2, 0.3])
    fwd = O.fused_bias_act_modes(x, b, None, 3, 0, 0.2, 2 ** 0.5)
    assert torch.equal(fwd, O.fused_bias_act(x, b))
    xr = x.clone().requires_grad_(True)
    go = torch.randn(2, 3, 4, 5, generator=torch.Generator().manual_seed(6))
    (gx,) = torch.autograd.grad(O.fused_bias_act(xr, b), xr, go)
    assert torch.allclose(O.fused_bias_act_modes(go, None, fwd, 3, 1, 0.2, 2 ** 0.5), gx, atol=1e-7)
    assert not O.fused_bias_act_modes(x, b, fwd, 3, 2, 0.2, 1.0).any() and not O.fused_bias_act_modes(x, b, fwd, 1, 2, 0.2, 1.0).any()
    assert torch.equal(O.fused_bias_act_modes(x, b, fwd, 1, 0, 0.2, 3.0), (x + b.view(1, 3, 1, 1)) * 3.0)


def test_oracle_upfirdn2d_matches_reference_native():
    """upfirdn2d_native is the only pure-torch restatement the reference itself ships (upfirdn2d.py:156-186)."""
    import importlib.util
    import sys
    import types
    src = open(os.path.join(ref_loader.REF, 'basicsr/ops/upfirdn2d/upfirdn2d.py')).read()
    start = src.index('def upfirdn2d_native')
    ns = {'F': torch.nn.functional, 'torch': torch}
    exec(compile(src[start:], 'upfirdn2d_native', 'exec'), ns)
    x = seeded_randn((2, 3, 9, 7), 6)
    k = seeded_randn((4, 4), 7)
    for up, down, pad in ((1, 1, (1, 2)), (2, 1, (2, 1)), (1, 2, (1, 1)), (2, 2, (0, 3)), (1, 1, (-1, 2))):
        ref = ns['upfirdn2d_native'](x, k, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        got = O.upfirdn2d(x, k, up=up, down=down, pad=pad)
        assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-6), (up, down, pad)


# ---- fixtures made from the reference's own input images (oracle/make_golden_real.py) ---------------------------------------
def _input_from_u8(img_bgr):
    """img2tensor(img/255., bgr2rgb=True, float32=True) + normalize(0.5, 0.5) (inference_codeformer.py:199-201)."""
    t = torch.from_numpy(np.ascontiguousarray((img_bgr[:, :, ::-1] / 255.).astype(np.float32).transpose(2, 0, 1)))
    return ((t - 0.5) / 0.5).unsqueeze(0)


def test_oracle_on_a_real_aligned_face_matches_reference_golden(seed0_net, golden_dir):
    g = np.load(os.path.join(golden_dir, 'real_0143.npz'))
    x = _input_from_u8(g['img'])
    out, logits, lq, idx = O.codeformer_forward(x, _sd(seed0_net), w=0.5, adain_flag=True, return_idx=True)
    assert float((logits - torch.from_numpy(g['logits'])).abs().max()) <= 2e-5
    safe = g['gap'].reshape(-1) >= 1e-5
    assert np.array_equal(idx.numpy().reshape(-1)[safe], g['idx'].reshape(-1)[safe])
    assert float((out[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs().max()) <= 1e-4
    u8 = O.tensor2img_u8(out[0])
    assert int(np.abs(u8.astype(np.int16) - g['out_u8'].astype(np.int16)).max()) <= 1 and (u8 != g['out_u8']).mean() < 1e-4


def test_oracle_vq_forward_statistics_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'vq_seed11.npz'))
    s = np.load(os.path.join(golden_dir, 'vq_stats_seed11.npz'))
    o = O.vq_forward(torch.from_numpy(g['z']), torch.from_numpy(g['codebook']), 0.25)
    assert np.array_equal(o['min_encoding_indices'].view(-1).numpy(), g['idx'])
    assert np.array_equal(o['min_encodings'].sum(0).numpy(), s['counts'])
    for k in ('loss', 'perplexity', 'mean_distance'):
        assert abs(float(o[k]) - float(s[k])) <= 1e-6 * abs(float(s[k])), k


def test_oracle_tensor2img_and_composite_match_reference_golden(golden_dir):
    k = np.load(os.path.join(golden_dir, 'tensor2img_kat.npz'))
    assert np.array_equal(O.tensor2img_u8(torch.from_numpy(k['t'])), k['img'])
    rep = json.load(open(os.path.join(golden_dir, 'real_oracle_vs_reference.json')))
    assert rep['00105.png']['oracle_composite_equal'] and rep['00105.png']['masked_pixels'] > 1000
    for name in ('0143.png', '0342.png', 'Solvay_conference_1927_0018.png'):
        assert rep[name]['oracle_out'] <= 1e-5 and rep[name]['idx_equal'] and rep[name]['oracle_u8_equal']
    # composite restatement on the stored mask: white input pixels take the network output, the rest keep the input
    g = np.load(os.path.join(golden_dir, 'real_masked_00105.npz'))
    x = _input_from_u8(g['img'])
    y = torch.zeros_like(x) - 0.25
    comp = O.inpaint_composite(x, y)
    m = torch.from_numpy(g['mask']).bool().expand_as(x)
    assert torch.equal(comp[m], y[m]) and torch.equal(comp[~m], x[~m])


@pytest.mark.parametrize('kind', ['big', 'small'])
def test_oracle_on_range_variants_matches_reference_golden(seed0_net, golden_dir, kind):
    """Range-robustness fixtures (oracle/make_golden_range.py): the oracle on weights whose un-normalised streams reach 1e14 / 5e-6
    reproduces the reference's logits, indices and (relative to the output scale) pixels; the recorded agreement is asserted too."""
    from oracle.synth import range_variant
    g = np.load(os.path.join(golden_dir, f'range_{kind}_seed.npz'))
    calib = {str(k): float(v) for k, v in zip(g['calib_keys'], g['calib_vals'])}
    sd = range_variant(_sd(seed0_net), kind, calib=calib or None)
    out, logits, lq, idx = O.codeformer_forward(seeded_input(1), sd, w=0.5, adain_flag=True, return_idx=True)
    assert float((logits - torch.from_numpy(g['logits'])).abs().max()) <= 2e-5
    assert np.array_equal(idx.numpy(), g['idx'])
    assert float((out[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs().max()) <= 1e-4 * max(1.0, float(g['out_absmax']))
    rep = json.load(open(os.path.join(golden_dir, 'range_oracle_vs_reference.json')))
    for name, r in rep.items():
        assert r['oracle_idx_equal'] and r['oracle_vs_ref_logits'] <= 2e-5 and r['oracle_vs_ref_out'] <= 1e-5, name
