"""-m gpu: bf16 STORAGE of activations (cf_conv_desc.io_bf16, ABI v22; precision 'bf16' = BASELINE configs 3 / 5: "bf16 storage + fp32
accumulate in generator + CFT").

Kernel level: a launch on bf16 tensors must be BITWISE the launch of the same kernel on the widened fp32 tensors with its output rounded
to bf16 once (widening is exact, the arithmetic in between is the same instantiation's, the GroupNorm partials are taken before the
rounding) -- for every kernel family the mode runs from 64x64 pixels up.  Network level: logits bitwise those of the default mode (encoder,
Transformer and argmax never see bf16), code indices exact, pixels inside the gate derived from the intrinsic cost of this arithmetic
(tools/bf16_gate_derivation.py), storage on vs off.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from codeformer_amd import lib, ops as _ops
    lib.load()
    return _ops


def _pair(ops, x, pw, *, x2=None, res=None, sft=None, **kw):
    """The launch on bf16 tensors and on their fp32 widenings; returns (bf16 output, fp32 output rounded to bf16, the two outputs' stats)."""
    import torch
    f = lambda t: None if t is None else t.float()
    y16 = ops.conv2d(x, pw, x2=x2, res=res, sft_scale=sft, **kw)
    y32 = ops.conv2d(f(x), pw, x2=f(x2), res=f(res), sft_scale=f(sft), **kw)
    assert y32.dtype == torch.float32
    return y16, y32


def test_to_bf16_is_round_to_nearest_even(ops):
    import torch
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 16, 64, generator=g) * 10 ** torch.randint(-6, 7, (2, 16, 16, 64), generator=g).float()
    x.view(-1)[:8] = torch.tensor([1.00390625, 1.01171875, -1.00390625, 0.0, -0.0, 3.3895313892515355e38, 1e-40, -2.5])   # ties both ways, max finite, subnormal
    xc = x.cuda()
    xc._cf_stats = 'tag'
    y = ops.to_bf16(xc)
    assert y.dtype == torch.bfloat16 and torch.equal(y.cpu(), x.to(torch.bfloat16)) and y._cf_stats == 'tag'
    assert ops.to_bf16(y) is y


@pytest.mark.parametrize('case', ['wino_swish_res', 'wino_cat', 'wino_leaky_sft', 'wino_none_act', 'direct64_swish', 'direct64_res', 'up2x',
                                  'conv1_cat', 'conv1_64', 'conv1s_cat', 'conv1s_64', 'head'])
def test_bf16_tensor_launch_is_the_fp32_tensor_launch_rounded_once(ops, case):
    import torch
    g = torch.Generator().manual_seed(hash(case) % 1000)
    rn = lambda *s: torch.randn(*s, generator=g)
    bf = lambda t: t.cuda().to(torch.bfloat16)
    H = 64
    kw, x2, res, sft, up, taps = {}, None, None, None, False, 3
    B, cin, cout, code = 2, 256, 256, 1
    if case == 'wino_swish_res':
        kw = dict(prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, emit_stats=True)
    elif case == 'wino_cat':
        B, cin, cout = 1, 256, 128
        kw = dict(prologue=ops.PRO_AFFINE_SWISH, emit_stats=True)
    elif case == 'wino_leaky_sft':
        B, cin, cout = 1, 128, 128
        kw = dict(prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, sft_w=0.7, emit_stats=True)
    elif case == 'wino_none_act':
        B, cin, cout = 1, 128, 128
        kw = dict(emit_stats=True)
    elif case == 'direct64_swish':
        B, cin, cout = 1, 128, 64
        kw = dict(prologue=ops.PRO_AFFINE_SWISH, emit_stats=True)
    elif case == 'direct64_res':
        B, cin, cout = 2, 64, 64
        kw = dict(prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, emit_stats=True)
    elif case == 'up2x':
        B, cin, cout, H, up = 1, 256, 256, 32, True
        kw = dict(upsample=True, emit_stats=True)
    elif case == 'conv1_cat':
        B, cin, cout, taps, code = 1, 512, 256, 1, 0
    elif case == 'conv1_64':
        B, cin, cout, taps, code = 2, 128, 64, 1, 0
    elif case == 'conv1s_cat':      # the streaming split-half 1x1 (what the network's skip convolutions run in the storage mode)
        B, cin, cout, taps, code = 1, 512, 256, 1, ops.SPLIT
    elif case == 'conv1s_64':
        B, cin, cout, taps, code = 2, 128, 64, 1, ops.SPLIT
    elif case == 'head':
        B, cin, cout, code = 1, 64, 3, 0
        kw = dict(prologue=ops.PRO_AFFINE, out_nchw=True)
    w = rn(cout, cin, taps, taps) * (2.0 / (taps * taps * cin)) ** 0.5
    b = rn(cout) * 0.1
    x = bf(rn(B, H, H, cin))
    if case in ('wino_cat', 'conv1_cat', 'conv1s_cat'):
        x, x2 = x[..., :cin // 2].contiguous(), x[..., cin // 2:].contiguous()
    Ho = 2 * H if up else H
    if kw.get('epilogue') in (ops.EPI_RESIDUAL, ops.EPI_SFT):
        res = bf(rn(B, Ho, Ho, cout))
    if kw.get('epilogue') == ops.EPI_SFT:
        sft = bf(rn(B, Ho, Ho, cout) * 0.3)
    if kw.get('prologue') in (ops.PRO_AFFINE, ops.PRO_AFFINE_SWISH):
        kw.update(scale=(torch.rand(B, cin, generator=g) + 0.5).cuda(), shift=(rn(B, cin) * 0.1).cuda())
    hw = (H, H)
    if taps == 3 and code == 1:
        code = ops.conv_code(1, cin, cout, H, H, up2x=up, c_split=None if x2 is None else x.shape[3])
        assert code == (ops.WBF16 if case.startswith('wino') else 1), (case, code)
    pw = ops.pack_weight(w.cuda(), b.cuda(), bf16=code, up2x=up)
    if ops.needs_act_scale(pw) and kw.get('prologue', ops.PRO_NONE) in (ops.PRO_NONE, ops.PRO_LEAKY):
        kw['act'] = ops.act_scale(x.float(), None if x2 is None else x2.float())
    y16, y32 = _pair(ops, x, pw, x2=x2, res=res, sft=sft, **kw)
    if case == 'head':
        assert y16.dtype == torch.float32 and torch.equal(y16, y32)        # (the NCHW image stays fp32)
        return
    assert y16.dtype == torch.bfloat16 and y16.shape == y32.shape
    assert torch.equal(y16, y32.to(torch.bfloat16)), (case, float((y16.float() - y32).abs().max()))
    if kw.get('emit_stats'):
        assert torch.equal(y16._cf_stats.part, y32._cf_stats.part)       # partials from the fp32 values, before the rounding
    # and it is the right convolution: fp64 reference on the same bf16-valued inputs, tolerance of bf16 operands + one output rounding
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], dim=3)
    xd = xin.double().cpu()
    if 'scale' in kw:
        xd = xd * kw['scale'].double().cpu()[:, None, None, :] + kw['shift'].double().cpu()[:, None, None, :]
        if kw['prologue'] == ops.PRO_AFFINE_SWISH:
            xd = xd * torch.sigmoid(xd)
    elif kw.get('prologue') == ops.PRO_LEAKY:
        xd = torch.nn.functional.leaky_relu(xd, 0.2)
    xd = xd.permute(0, 3, 1, 2)
    if up:
        xd = torch.nn.functional.interpolate(xd, scale_factor=2.0, mode='nearest')
    ref = torch.nn.functional.conv2d(xd, w.double(), b.double(), padding=taps // 2).permute(0, 2, 3, 1)
    if kw.get('epilogue') == ops.EPI_RESIDUAL:
        ref = ref + res.double().cpu()
    elif kw.get('epilogue') == ops.EPI_SFT:
        ref = res.double().cpu() + 0.7 * (res.double().cpu() * sft.double().cpu() + ref)
    err = float((y16.double().cpu() - ref).abs().max())
    assert err <= (0.06 if code else 0.03) * max(1.0, float(ref.abs().max()) / 4), (case, err)


def test_bf16_tensors_are_refused_by_kernels_without_the_storage_form(ops):
    import torch
    x = torch.randn(1, 64, 64, 128, device='cuda').to(torch.bfloat16)
    w = torch.randn(128, 128, 3, 3, device='cuda') * 0.03
    for code in (ops.WSPLIT, ops.WF43, ops.WINOGRAD, ops.SPLIT, 0):    # (3x3: the split-half 1x1 streaming form does take bf16 tensors)
        pw = ops.pack_weight(w, None, bf16=code)
        with pytest.raises(RuntimeError, match='io_bf16'):
            ops.conv2d(x, pw, act=ops.act_scale(x.float()) if ops.needs_act_scale(pw) else None)
    with pytest.raises(TypeError):   # mixed storage types in one launch
        ops.conv2d(x, ops.pack_weight(w, None, bf16=ops.WBF16), epilogue=ops.EPI_RESIDUAL, res=torch.zeros(1, 64, 64, 128, device='cuda'))


@pytest.fixture(scope='module')
def net():
    import importlib.util
    spec = importlib.util.spec_from_file_location('gpu_check', os.path.join(ROOT, 'tools', 'gpu_check.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.build_net().cuda()


def test_network_with_bf16_storage(net):
    """precision 'bf16' with storage (the default since round 6) and without: logits bitwise those of the default mode, indices exact, pixels
    inside the derived gate; the storage form moves the picture by less than the gate's distance from the operand-only form; batch invariant."""
    import torch
    from oracle.synth import seeded_input
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'restoration_seed0_face0_w0.7.npz'))
    g0 = np.load(os.path.join(ROOT, 'tests', 'golden', 'restoration_seed0_face0.npz'))
    x = seeded_input(2).cuda()
    net.precision = 'f16x2'
    _, logits_default, _ = net(x[:1], w=0.7, adain=True)
    outs = {}
    for storage in (True, False):
        net.precision, net.bf16_storage = 'bf16', storage
        out, logits, _ = net(x[:1], w=0.7, adain=True)
        assert out.dtype == torch.float32 and torch.equal(logits, logits_default)
        assert np.array_equal(net.last_indices.cpu().numpy().reshape(-1), g0['idx'].reshape(-1))
        d = (out.cpu()[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs()
        print(f'bf16 mode, storage {storage}: max|d| {float(d.max()):.4f} mean|d| {float(d.mean()):.5f}')
        assert float(d.max()) <= 0.196 and float(d.mean()) <= 0.0152
        outs[storage] = out
    net.bf16_storage = True
    two = net(x, w=0.7, adain=True)[0]
    assert torch.equal(two[:1], outs[True])          # batch invariance holds in the storage mode
    again = net(x[:1], w=0.7, adain=True)[0]
    assert torch.equal(again, outs[True])            # run to run
    net.precision = 'f16x2'
