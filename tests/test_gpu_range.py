"""-m gpu: range robustness of the 16-bit-operand kernels.

tests/golden/range_*.npz hold the REFERENCE's CodeFormer.forward (oracle/make_golden_range.py) for weights that push the
un-normalised streams -- residual stream into Upsample.conv, quantised feature, the CFT branch -- to 1e3..1e14 ('big'), down to
1e-4..5e-6 ('small'), and for trained-like heavy-tailed weights ('heavy', seeded face and the reference's own crop 0143.png).
The HIP path must meet the north-star gates in the default split-half mode AND the exact mode:
  logits 1e-4, code indices exact (reference top-2 gap >= 1e-5), pixels 1e-3 x max(1, max |reference output|).
Without the per-image power-of-two range scale (cf_conv_desc.act_scale) the 'big' case is inf / NaN and the 'small' case loses the
lo halves; the last test measures that (informational print + the assertion that the scale is what fixes it).
"""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
pytestmark = pytest.mark.gpu

CASES = [('big', 'seed'), ('small', 'seed'), ('heavy', 'seed'), ('heavy', 'real0143')]


@pytest.fixture(scope='module')
def chk():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from codeformer_amd import lib
    lib.load()
    spec = importlib.util.spec_from_file_location('gpu_check', os.path.join(ROOT, 'tools', 'gpu_check.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope='module')
def nets(chk):
    """One module per variant (weights = range_variant of the seed-0 state_dict; 'big' uses the golden's calibration table)."""
    from oracle.synth import range_variant
    base = chk.build_net()
    sd0 = {k: v.detach().clone() for k, v in base.state_dict().items()}
    cache = {}

    def get(kind):
        if kind not in cache:
            g = np.load(os.path.join(GOLD, f'range_{kind}_seed.npz'))
            calib = {str(k): float(v) for k, v in zip(g['calib_keys'], g['calib_vals'])}
            net = chk.build_net()
            net.load_state_dict(range_variant(sd0, kind, calib=calib or None))
            cache[kind] = net.cuda()
        return cache[kind]
    return get


def _input(tag):
    import torch
    from oracle.synth import seeded_input
    if tag == 'seed':
        return seeded_input(1).cuda()
    from codeformer_amd import ops
    img = np.load(os.path.join(GOLD, 'real_0143.npz'))['img']
    return ops.img_u8_to_tensor(torch.from_numpy(img).unsqueeze(0).cuda())


def _run(net, x, precision):
    import torch
    net.precision = precision
    try:
        out, logits, lq = net(x, w=0.5, adain=True)
        torch.cuda.synchronize()
    finally:
        net.precision = 'f16x2'
    return out.cpu(), logits.cpu(), lq.cpu(), net.last_indices.cpu().numpy()


@pytest.mark.parametrize('precision', ['f16x2', 'fp32'])
@pytest.mark.parametrize('kind,tag', CASES)
def test_range_variants_match_the_reference(nets, kind, tag, precision):
    import torch
    g = np.load(os.path.join(GOLD, f'range_{kind}_{tag}.npz'))
    out, logits, lq, idx = _run(nets(kind), _input(tag), precision)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(logits).all())
    dl = float((logits - torch.from_numpy(g['logits'])).abs().max())
    scale = max(1.0, float(g['out_absmax']))
    dp = float((out[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs().max())
    dq = float((lq[:, ::8] - torch.from_numpy(g['lq_sub'])).abs().max()) / max(float(g['lq_absmax']), 1e-30)
    ref, gap = g['idx'].reshape(-1), g['gap'].reshape(-1)
    safe = gap >= 1e-5
    nbad = int((idx.reshape(-1)[safe] != ref[safe]).sum())
    print(f'range {kind}/{tag} [{precision}]: logits {dl:.2e}  pixels {dp:.2e} (output scale {scale:.2f})  lq_feat rel {dq:.2e}  '
          f'indices differing {nbad}  near-ties {int((~safe).sum())}  min gap {float(gap.min()):.2e}')
    assert dl <= 1e-4
    assert nbad == 0
    assert dp <= 1e-3 * scale


def test_act_scale_kernels():
    """cf_act_scale_from_tensor: exact power of two with 4 * max|x| * s in [2^13, 2^14); from_stats: the same from the statistics
    partials of a producing conv -- a rigorous bound (>= the true maximum, loose by < 2^6); zero / non-finite images give s = 1."""
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 16, 16, 64, generator=g)
    x[0] *= 3.7e9
    x[1] *= 2.2e-7
    x[2] = 0
    x[3, 1, 2, 3] = float('inf')
    x[4, 0, 0, 0] = 12345.678
    xd = x.cuda()
    act = ops.act_scale(xd).cpu()
    for b in (0, 1, 4, 5):
        m = 4.0 * float(x[b].abs().max())
        s, inv = float(act[b, 0]), float(act[b, 1])
        assert s * inv == 1.0 and np.log2(s) == int(np.log2(s))
        assert 2.0 ** 13 <= m * s < 2.0 ** 14, (b, m, s)
    assert act[2].tolist() == [1.0, 1.0] and act[3].tolist() == [1.0, 1.0]
    # statistics route: producer = a 3x3 conv with emit_stats
    w = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    xin = torch.randn(2, 32, 32, 64, generator=g).cuda()
    xin[1] *= 1.0e6
    y = ops.conv2d(xin, ops.pack_weight(w.cuda(), None), emit_stats=True)
    a_stats = ops.act_scale(y).cpu()
    for b in range(2):
        m = 4.0 * float(y[b].abs().max())
        s = float(a_stats[b, 0])
        assert m * s < 2.0 ** 14 and m * s >= 2.0 ** 7, (b, m * s)       # never above the range, at most 6 bits below the tight scale
    # two tensors (the halves of a concatenated input) share one table: the scale of the larger bound; repeated launches reuse the cells
    y2 = ops.conv2d(xin * 37.0, ops.pack_weight(w.cuda(), None), emit_stats=True)
    for _ in range(3):
        pair = ops.act_scale(y, y2).cpu()
    a2 = ops.act_scale(y2).cpu()
    assert torch.equal(pair[:, 0], torch.minimum(a_stats[:, 0], a2[:, 0])) and torch.equal(pair[:, 1], torch.maximum(a_stats[:, 1], a2[:, 1]))
    assert not ops._act_cells(xd.device, 2).any()
    # the two-launch entry points of ABI v16 give the same tables
    from codeformer_amd import lib as L
    lib = L.load()
    old = torch.empty(2, 2, device='cuda')
    scratch = torch.empty(64, device='cuda')
    st = y._cf_stats
    L.check(lib.cf_act_scale_from_stats(L.ptr(st.part, dtype=torch.float64), 2, st.part.numel() // 4, 4.0, L.ptr(scratch), L.ptr(old), L.stream_ptr()), 'from_stats')
    assert torch.equal(old.cpu(), a_stats)
    old6 = torch.empty(6, 2, device='cuda')
    scratch = torch.empty(6 * 32, device='cuda')
    L.check(lib.cf_act_scale_from_tensor(L.ptr(xd), 6, xd.numel() // 6, 4.0, L.ptr(scratch), L.ptr(old6), L.stream_ptr()), 'from_tensor')
    assert torch.equal(old6.cpu(), act)


@pytest.mark.parametrize('mag', [1.0e9, 3.0e-9])
@pytest.mark.parametrize('pro', ['none', 'leaky'])
def test_split_conv_kernels_with_extreme_inputs(pro, mag):
    """The eight-wave Winograd kernel and the folded-upsample direct kernel on un-normalised inputs far outside the IEEE-half range,
    against fp64: with the range scale the error relative to the output scale is that of O(1) inputs."""
    import torch
    import torch.nn.functional as F
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, 128, 32, 32, generator=g)
    x[0] *= mag
    x[1] *= mag * 37.0
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    bias = torch.randn(128, generator=g) * 0.1
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    for up in (False, True):
        if up and pro == 'leaky':
            continue
        code = ops.conv_code(ops.SPLIT, 128, 128, 32, 32, up2x=up)
        pw = ops.pack_weight(w.cuda(), bias.cuda(), bf16=code, up2x=up)
        assert ops.needs_act_scale(pw)
        y = ops.conv2d(xn, pw, upsample=up, prologue=ops.PRO_LEAKY if pro == 'leaky' else ops.PRO_NONE, act=ops.act_scale(xn))
        xr = x.double()
        if pro == 'leaky':
            xr = F.leaky_relu(xr, 0.2)
        if up:
            xr = F.interpolate(xr, scale_factor=2.0, mode='nearest')
        ref = F.conv2d(xr, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        for b in range(2):
            err = float((y[b].cpu().double() - ref[b]).abs().max()) / float(ref[b].abs().max())
            print(f'split conv up={up} pro={pro} mag={mag:g} image {b}: max err / max|ref| = {err:.2e}')
            assert err < 3e-6


def test_without_the_range_scale_the_big_case_overflows(nets):
    """Documents what the scale is for: the same 'big' weights with CODEFORMER_HIP_RANGE_SCALE off give a non-finite image."""
    import torch
    from codeformer_amd import ops
    g = np.load(os.path.join(GOLD, 'range_big_seed.npz'))
    net = nets('big')
    x = _input('seed')
    ops.RANGE_SCALE = False
    try:
        out = net(x, w=0.5, adain=True)[0].cpu()
    finally:
        ops.RANGE_SCALE = True
    finite = bool(torch.isfinite(out).all())
    d = float((out[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs().nan_to_num(float('inf')).max())
    print(f'big variant without the range scale: finite={finite} pixels {d:.2e}')
    assert (not finite) or d > 1e-3


def test_groupnorm_range_fallback_to_exact_kernels(chk):
    """GroupNorm gains so large that |gamma| * sqrt(n - 1) + |beta| could leave the IEEE-half range: the arch modules must route those
    layers to the exact fp32 kernels (HipModule._range_code) -- the default mode then still matches the CPU oracle (which is pinned to
    the reference) at the north-star gates, relative to the output scale."""
    import torch
    from codeformer_amd import ops
    from oracle import codeformer_oracle as O
    from oracle.synth import seeded_input
    net = chk.build_net()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k in sd:
        # generator + CFT only: with such gains in the ENCODER the logits themselves are ill-conditioned (any two fp32 evaluations differ
        # by ~1e-3 there -- measured with this very test), which would test the conditioning of the weights, not the fallback
        if ('norm' in k or k.endswith('.23.weight')) and k.endswith('weight') and sd[k].dim() == 1 and k.startswith(('generator.', 'fuse_convs_dict.')):
            sd[k] = sd[k] * 60.0            # GroupNorm gains of 60: bound 60 * sqrt(n) >> 16376 on every level
    net.load_state_dict(sd)
    assert not ops.gn_range_ok(60.0, 0.0, 4 * 256 * 256)
    net = net.cuda()
    x = seeded_input(1)
    out, logits, _ = net(x.cuda(), w=0.5, adain=True)
    torch.cuda.synchronize()
    o_out, o_logits, _, o_idx = O.codeformer_forward(x, sd, w=0.5, adain_flag=True, return_idx=True)
    scale = max(1.0, float(o_out.abs().max()))
    dp, dl = float((out.cpu() - o_out).abs().max()), float((logits.cpu() - o_logits).abs().max())
    gap = torch.topk(o_logits, 2, dim=-1).values
    safe = ((gap[..., 0] - gap[..., 1]) >= 1e-5).view(-1)
    nbad = int((net.last_indices.cpu().view(-1)[safe] != o_idx.view(-1)[safe]).sum())
    print(f'GroupNorm gains x60 (exact-kernel fallback): pixels {dp:.2e} (output scale {scale:.2f}) logits {dl:.2e} indices differing {nbad}')
    assert bool(torch.isfinite(out).all()) and dp <= 1e-3 * scale and dl <= 1e-4 * max(1.0, float(o_logits.abs().max())) and nbad == 0


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_single_operand_modes_survive_the_big_streams(nets, precision):
    """The opt-in 16-bit operand modes on the 'big' variant: the range scale (and, for single IEEE halves on un-normalised inputs, the
    switch to the split kernels) keeps them finite and inside their stated pixel gates relative to the output scale."""
    import torch
    g = np.load(os.path.join(GOLD, 'range_big_seed.npz'))
    out, logits, _, idx = _run(nets('big'), _input('seed'), precision)
    scale = max(1.0, float(g['out_absmax']))
    d = (out[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs()
    gmax = {'fp16': 0.04, 'bf16': 0.196}[precision]
    print(f'range big [{precision}]: finite {bool(torch.isfinite(out).all())} max|d| {float(d.max()):.3e} mean|d| {float(d.mean()):.3e} (output scale {scale:.2f})')
    assert bool(torch.isfinite(out).all()) and float(d.max()) <= gmax * scale
    assert np.array_equal(idx.reshape(-1), g['idx'].reshape(-1))      # the encoder runs on split halves in these modes: indices as the default mode's
