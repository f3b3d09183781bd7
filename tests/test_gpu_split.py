"""-m gpu: the split-half convolution kernel (cf_split.hip, CF_OPERAND_F16X2: fp32 operands as hi + lo IEEE halves, three f16
MFMAs per product) through the C ABI against fp64 references -- every prologue / epilogue / concat / upsample combination the
generator and the fusion blocks use, extreme weight / activation magnitudes, the epilogue's GroupNorm partials, bitwise
repeatability and batch invariance, and the refusals of the C ABI for shapes the kernel does not cover.

Tolerance: 2e-5 + 1e-5*|ref| (the bound of the exact-fp32 kernels' tests); additionally the split kernel's max error must stay
within 5x of the exact-fp32 kernel's on the same case (measured: 1-2.3x max, equal mean, on the layers it serves; 4.4x on a
512-channel 16x16 layer, which the network gives to the Winograd form).  The Winograd form of the split-half scheme
(cf_winograd.hip H2, cf_wsplit.hip) is held to the same bounds (measured: 0.5-0.9x of the exact Winograd kernel's error).
"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def sc():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from codeformer_amd import lib
    lib.load()
    spec = importlib.util.spec_from_file_location('split_check', os.path.join(ROOT, 'tools', 'split_check.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_split_conv_against_fp64(sc):
    import math
    nw = 0
    for c in sc.CASES:
        es, ef, est, rmax, ems, ew, estw = sc.case(timing=False, **c)
        assert es <= 2e-5 + 1e-5 * rmax, (c, es)
        assert es <= 5.0 * ef + 1e-7 * rmax, (c, es, ef)
        assert est <= 1e-4, (c, est)
        if not math.isnan(ew):          # the Winograd form of the split-half scheme (cf_winograd.hip H2 / cf_wsplit.hip)
            nw += 1
            assert ew <= 2e-5 + 1e-5 * rmax and ew <= 4.0 * ef + 1e-7 * rmax and estw <= 1e-4, (c, ew, ef, estw)
    assert nw >= 8


def test_split_conv_is_bitwise_repeatable_and_batch_invariant(sc):
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 32, 48, 128, generator=g).cuda()
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda()
    b = torch.randn(128, generator=g).cuda()
    for up in (False, True):
        pw = ops.pack_weight(w, b, bf16=ops.SPLIT, up2x=up)
        y = ops.conv2d(x, pw, upsample=up, emit_stats=True)
        y2 = ops.conv2d(x, pw, upsample=up, emit_stats=True)
        assert torch.equal(y, y2) and torch.equal(y._cf_stats.part, y2._cf_stats.part)
        y1 = ops.conv2d(x[1:2].contiguous(), pw, upsample=up, emit_stats=True)
        assert torch.equal(y[1:2], y1)
        assert torch.equal(y._cf_stats.part.view(3, -1)[1:2], y1._cf_stats.part.view(1, -1))   # (the partials too: tile widths are per-image)
    # Winograd form: the eight-wave kernel (128 channels), the four-wave kernel (64 channels) and its split-K instantiation (16x16)
    for cin, cout, h, wd in ((128, 128, 32, 48), (64, 64, 32, 48), (128, 128, 16, 16)):
        xx = torch.randn(3, h, wd, cin, generator=g).cuda()
        pw = ops.pack_weight((torch.randn(cout, cin, 3, 3, generator=g) * 0.03).cuda(), b[:cout].contiguous(), bf16=ops.WSPLIT)
        y = ops.conv2d(xx, pw, emit_stats=True)
        y2 = ops.conv2d(xx, pw, emit_stats=True)
        assert torch.equal(y, y2) and torch.equal(y._cf_stats.part, y2._cf_stats.part)
        y1 = ops.conv2d(xx[1:2].contiguous(), pw, emit_stats=True)
        assert torch.equal(y[1:2], y1) and torch.equal(y._cf_stats.part.view(3, -1)[1:2], y1._cf_stats.part.view(1, -1))
    # a single 16-channel slab (pipeline shorter than its depth) through the eight-wave kernel, against the exact Winograd kernel
    xx = torch.randn(2, 32, 32, 16, generator=g).cuda()
    w16 = (torch.randn(128, 16, 3, 3, generator=g) * 0.1).cuda()
    yw = ops.conv2d(xx, ops.pack_weight(w16, b, bf16=ops.WSPLIT))
    yf = ops.conv2d(xx, ops.pack_weight(w16, b, bf16=ops.WINOGRAD))
    assert float((yw - yf).abs().max()) <= 1e-5


def test_split_conv_refusals_and_overflow_is_loud(sc):
    import torch
    from codeformer_amd import ops
    w = torch.zeros(64, 64, 3, 3, device='cuda')
    w[:, :, 1, 1] = torch.eye(64, device='cuda')
    pw = ops.pack_weight(w, None, bf16=ops.SPLIT)
    x = torch.randn(1, 16, 16, 64, device='cuda')
    y = ops.conv2d(x, pw)
    assert float((y - x).abs().max()) <= 2e-7 * float(x.abs().max())      # identity kernel: only the 22-bit operand split is visible
    with pytest.raises(RuntimeError, match='f16x2'):
        ops.conv2d(torch.zeros(1, 20, 16, 64, device='cuda'), pw)            # 20 rows: not a whole number of 8x16 tiles
    with pytest.raises(ValueError, match='stride2'):
        ops.conv2d(x, pw, stride=2)                                          # stride 2 has its own packed form (pack_weight(stride2=True))
    with pytest.raises(ValueError):
        ops.pack_weight(torch.zeros(64, 48, 3, 3, device='cuda'), None, bf16=ops.SPLIT)   # cin % 32
    # activations beyond the IEEE-half range (|x| > 65504) do not produce a silently wrong finite value
    x2 = x.clone()
    x2[0, 3, 3, 5] = 1e6
    assert not torch.isfinite(ops.conv2d(x2, pw)).all()
    # host policy: which layers take the split kernel
    assert ops.SPLIT_WINOGRAD     # (default; CODEFORMER_HIP_SPLIT_WINOGRAD=0 keeps eligible layers on the direct split-half kernel)
    assert ops.conv_code(ops.SPLIT, 128, 128, 256, 256) == ops.WSPLIT and ops.conv_code(ops.SPLIT, 512, 512, 16, 16) == ops.WSPLIT
    assert ops.conv_code(ops.SPLIT_DIRECT, 512, 512, 16, 16) == 0 and ops.conv_code(ops.SPLIT, 48, 64, 64, 64) == ops.WSPLIT
    assert ops.conv_code(ops.SPLIT, 512, 512, 16, 16, up2x=True) == ops.SPLIT and ops.conv_code(ops.SPLIT, 128, 128, 256, 256, up2x=True) == ops.SPLIT


def test_split_half_token_gemm(sc):
    """cf_gemm_split.hip (CodeFormer.gemm_precision = 'f16x2', off by default): against fp64 at least as close as the exact fp32 GEMM
    (bias / GELU / residual epilogues, extreme weight magnitudes), bitwise the same for split counts 1 / 2 / 4 / 8, and the network's
    logits and indices with it against the reference golden."""
    import importlib.util
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location('gemm_split_check', os.path.join(ROOT, 'tools', 'gemm_split_check.py'))
    gc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gc)
    for c in gc.CASES:
        es, ef, rmax, same = gc.case(**c)
        assert same and es <= 2e-5 + 1e-5 * rmax and es <= 2.0 * ef + 1e-7 * rmax, (c, es, ef)
    spec = importlib.util.spec_from_file_location('gpu_check', os.path.join(ROOT, 'tools', 'gpu_check.py'))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    from oracle.synth import seeded_input
    net = chk.build_net().cuda()
    net.gemm_precision = 'f16x2'
    g = np.load(os.path.join(ROOT, 'tests/golden/restoration_seed0_face0.npz'))
    x = seeded_input(2).cuda()
    out, logits, _ = net(x, w=0.5, adain=True)
    assert float((logits[:1].cpu() - torch.from_numpy(g['logits'])).abs().max()) <= 1e-4
    assert np.array_equal(net.last_indices[:1].cpu().numpy(), g['idx'])
    out1, logits1, _ = net(x[:1].contiguous(), w=0.5, adain=True)
    assert torch.equal(logits1, logits[:1]) and torch.equal(out1, out[:1])      # batch invariance holds with it


def test_splitk_gemm_bits_do_not_depend_on_the_split_count(sc):
    """1x1 / Linear on small token images (cf_conv_desc.split_k): K is always cut into virtual chunks of 128 added in a fixed order,
    so the result is bitwise the same whether 1, 2, 4 or 8 workgroups share a tile -- and equals the fp64 product to fp32 accuracy."""
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(5)
    for (M, K, N, epi) in ((256, 512, 512, ops.EPI_RESIDUAL), (512, 1024, 512, ops.EPI_GELU), (256, 512, 1536, ops.EPI_NONE), (1024, 256, 128, ops.EPI_NONE)):
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        pw = ops.pack_weight(w, b)
        outs = []
        v = K // 128
        for ns in [n for n in (1, 2, 4, 8) if v % n == 0]:
            y = ops.conv2d(x.view(M // 256, 16, 16, K), pw, epilogue=epi, res=res.view(M // 256, 16, 16, N) if epi == ops.EPI_RESIDUAL else None,
                           split_k=ns, emit_stats=epi == ops.EPI_RESIDUAL)
            outs.append(y.view(M, N))
            again = ops.conv2d(x.view(M // 256, 16, 16, K), pw, epilogue=epi, res=res.view(M // 256, 16, 16, N) if epi == ops.EPI_RESIDUAL else None,
                               split_k=ns)
            assert torch.equal(again.view(M, N), outs[-1])
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), (M, K, N)
        ref = x.double() @ w.double().t() + b.double()
        if epi == ops.EPI_RESIDUAL:
            ref = ref + res.double()
        elif epi == ops.EPI_GELU:
            ref = torch.nn.functional.gelu(ref)
        assert float((outs[0].double() - ref).abs().max()) <= 2e-5 + 1e-5 * float(ref.abs().max())
    # policy: eligibility by per-image shape; the count by what is in flight
    pw = ops.pack_weight(torch.zeros(512, 512, device='cuda'))
    assert ops.splitk_for(pw, 16, 16, 512, 1) == 4 and ops.splitk_for(pw, 16, 16, 512, 16) == 1 and ops.splitk_for(pw, 64, 64, 512, 1) == 0


def test_fp32_token_tile_gemm_is_bitwise_the_splitk_gemm(sc):
    """Round 6: fp32 Linear launches with one workgroup per tile on at least 128 tokens run gemm_f32_tile_kernel (cf_gemm_split.hip); its
    bits must be those of the 64x64 split-K instantiation (any split count), whole and per image of the batch -- the host picks by batch."""
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(11)
    for (B, K, N, epi) in ((16, 512, 512, ops.EPI_RESIDUAL), (16, 512, 1024, ops.EPI_GELU), (8, 1024, 512, ops.EPI_NONE), (16, 256, 512, ops.EPI_NONE), (2, 512, 1024, ops.EPI_NONE)):
        x = torch.randn(B, 16, 16, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        res = torch.randn(B, 16, 16, N, generator=g).cuda() if epi == ops.EPI_RESIDUAL else None
        pw = ops.pack_weight(w, b)
        y1 = ops.conv2d(x, pw, epilogue=epi, res=res, split_k=1)                 # the tile kernel (M % 128 == 0)
        for ns in (2, 4):
            if (K // 128) % ns == 0:
                assert torch.equal(y1, ops.conv2d(x, pw, epilogue=epi, res=res, split_k=ns)), (B, K, N, ns)
        one = ops.conv2d(x[:1].contiguous(), pw, epilogue=epi, res=None if res is None else res[:1].contiguous())   # (host's choice for one face: a split)
        assert torch.equal(one, y1[:1])
        assert torch.equal(y1, ops.conv2d(x, pw, epilogue=epi, res=res))      # the host's own choice at this batch
        ref = x.double().view(-1, K) @ w.double().t() + b.double()
        if epi == ops.EPI_RESIDUAL:
            ref = ref + res.double().view(-1, N)
        elif epi == ops.EPI_GELU:
            ref = torch.nn.functional.gelu(ref)
        assert float((y1.double().view(-1, N) - ref).abs().max()) <= 2e-5 + 1e-5 * float(ref.abs().max())


def test_two_token_matrices_in_one_split_half_gemm(sc):
    """cf_conv_desc.in0_alt (round 6): the columns >= alt_from of a split-half token GEMM contract a second token matrix -- q | k on LN(x) + pos
    and v on LN(x) as ONE launch.  Bitwise the two single-matrix launches with the same packed weight, in every kernel the host may pick
    (in-workgroup split: few tokens; token tiles: many; cross-workgroup split), and refused where it does not belong."""
    import pytest
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(21)
    E = 512
    w = (torch.randn(3 * E, E, generator=g) / E ** 0.5).cuda()
    b = torch.randn(3 * E, generator=g).cuda()
    pw = ops.pack_weight(w, b, bf16=ops.GSPLIT)
    for B, sk in ((1, None), (16, None), (2, 2), (8, None)):
        xa = torch.randn(B * 256, E, generator=g).cuda()
        xb = torch.randn(B * 256, E, generator=g).cuda()
        v4 = lambda t: t.view(B, 16, 16, E)
        both = ops.conv2d(v4(xa), pw, x_alt=v4(xb), alt_from=2 * E, split_k=sk).view(B * 256, 3 * E)
        ya = ops.conv2d(v4(xa), pw, split_k=sk).view(B * 256, 3 * E)
        yb = ops.conv2d(v4(xb), pw, split_k=sk).view(B * 256, 3 * E)
        assert torch.equal(both[:, :2 * E], ya[:, :2 * E]) and torch.equal(both[:, 2 * E:], yb[:, 2 * E:]), (B, sk)
        ref = torch.cat([xa.double() @ w[:2 * E].double().t(), xb.double() @ w[2 * E:].double().t()], dim=1) + b.double()
        assert float((both.double() - ref).abs().max()) <= 2e-5 + 1e-5 * float(ref.abs().max())
    with pytest.raises((RuntimeError, ValueError)):
        ops.conv2d(xa.view(8, 16, 16, E), ops.pack_weight(w, b), x_alt=xb.view(8, 16, 16, E), alt_from=2 * E)      # fp32 operands: no such form
    with pytest.raises(RuntimeError):
        ops.conv2d(xa.view(8, 16, 16, E), pw, x_alt=xb.view(8, 16, 16, E), alt_from=100)                            # not a multiple of 128


def test_fused_finalize_is_bitwise_the_separate_launches(sc):
    """cf_groupnorm_finalize2 (round 6): GroupNorm tables of one tensor or of a concatenated pair, and the range-scale table of the same
    tensor(s), in one launch -- bitwise the tables of cf_groupnorm_finalize per tensor + cf_act_scale_fused (ops.FINALIZE_FUSED = False)."""
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(31)
    B = 3

    def producer(cin, cout, H, scale):
        x = (torch.randn(B, H, H, cin, generator=g) * scale).cuda()
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
        return ops.conv2d(x, ops.pack_weight(w, torch.randn(cout, generator=g).cuda(), bf16=ops.WINOGRAD), emit_stats=True)

    for (ca, cb, H, sa, sb) in ((128, 0, 32, 1.0, 0.0), (256, 256, 32, 3.0, 1e-3), (128, 128, 64, 1e4, 2.0), (64, 0, 64, 1e-6, 0.0)):
        ya = producer(64, ca, H, sa)
        yb = producer(64, cb, H, sb) if cb else None
        xs = [ya] if yb is None else [ya, yb]
        gamma = (torch.rand(ca + cb, generator=g) + 0.5).cuda()
        beta = torch.randn(ca + cb, generator=g).cuda()
        old = ops.FINALIZE_FUSED
        try:
            ops.FINALIZE_FUSED = False
            for t in xs:
                t.__dict__.pop('_cf_act', None)
                t.__dict__.pop('_cf_act_pair', None)
            sc0, sh0 = ops.groupnorm_tables(xs, gamma, beta, act_growth=4.0)
            act0 = ops.act_scale(ya, yb)
            for t in xs:
                t.__dict__.pop('_cf_act', None)
            ops.FINALIZE_FUSED = True
            sc1, sh1 = ops.groupnorm_tables(xs, gamma, beta, act_growth=4.0)
            parked = getattr(ya, '_cf_act_pair' if yb is not None else '_cf_act', None)
            assert parked is not None                            # the table came with the finalize launch ...
            act1 = ops.act_scale(ya, yb)
            assert act1 is parked[-1]                              # ... and act_scale found it instead of launching
            sc2, sh2 = ops.groupnorm_tables(xs, gamma, beta)      # tables only
        finally:
            ops.FINALIZE_FUSED = old
        assert torch.equal(sc0, sc1) and torch.equal(sh0, sh1) and torch.equal(sc0, sc2) and torch.equal(sh0, sh2), (ca, cb, H)
        assert torch.equal(act0, act1), (act0, act1)
        # and they are the right statistics
        full = torch.cat([t.double() for t in xs], dim=3).view(B, H * H, 32, (ca + cb) // 32)
        mean, var = full.mean((1, 3)), full.var((1, 3), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-6)
        want = (rstd[:, :, None] * gamma.double().view(32, -1)[None]).view(B, -1)
        assert float(((sc1.double() - want).abs() / want.abs().clamp_min(1e-12)).max()) < 1e-5


def test_splitk_winograd_bits_do_not_depend_on_the_split_count(sc):
    """The Winograd kernel on images of at most 32x32 pixels: virtual chunks of 128 channels are taken to the output domain and added
    in a fixed order -- 1, 2 or 4 workgroups per patch give the same bits; errors vs fp64 as the unsplit kernel's."""
    import torch
    import torch.nn.functional as F
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(9)
    for (B, H, W, cin, cout, epi) in ((1, 16, 16, 512, 512, ops.EPI_RESIDUAL), (2, 32, 32, 256, 256, ops.EPI_NONE), (1, 16, 16, 256, 512, ops.EPI_NONE)):
        x = torch.randn(B, H, W, cin, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        sc_, sh_ = torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g) * 0.1
        res = torch.randn(B, H, W, cout, generator=g)
        pw = ops.pack_weight(w.cuda(), b.cuda(), bf16=ops.WINOGRAD)
        kw = dict(prologue=ops.PRO_AFFINE_SWISH, scale=sc_.cuda(), shift=sh_.cuda(), epilogue=epi, res=res.cuda() if epi else None)
        outs, stats = [], []
        for ns in [n for n in (1, 2, 4) if (cin // 128) % n == 0]:
            y = ops.conv2d(x.cuda(), pw, split_k=ns, emit_stats=True, **kw)
            outs.append(y)
            stats.append(y._cf_stats.part.clone())
        assert all(torch.equal(o, outs[0]) for o in outs[1:]) and all(torch.equal(t, stats[0]) for t in stats[1:]), (H, cin, cout)
        xd = x.double() * sc_.double()[:, None, None, :] + sh_.double()[:, None, None, :]
        xd = xd * torch.sigmoid(xd)
        ref = F.conv2d(xd.permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        if epi:
            ref = ref + res.double()
        assert float((outs[0].cpu().double() - ref).abs().max()) <= 2e-5 + 1e-5 * float(ref.abs().max())
        # one image of the batch alone: same bits (the split count follows the tiles in flight, the association does not)
        if B > 1:
            kw1 = dict(kw, scale=sc_[:1].cuda(), shift=sh_[:1].cuda(), res=res[:1].cuda() if epi else None)
            assert torch.equal(ops.conv2d(x[:1].cuda(), pw, **kw1), ops.conv2d(x.cuda(), pw, **kw)[:1])


def test_winograd_single_16bit_operands(sc):
    """precision 'fp16' / 'bf16' on the eight-wave Winograd kernel (ops.WF16 / ops.WBF16: U and V rounded once to 16 bits, one MFMA per
    transform-domain product).  Oracle: the SAME convolution in fp64 on operands rounded the way the kernel rounds them is not
    available in closed form (V is rounded after the input transform), so the gates are the operand formats' own error levels against
    the fp64 convolution of the unrounded operands -- K = 9*128 products of relative error 2^-11 (half) / 2^-8 (bf16) each -- and, as
    a sharper check of the data path, agreement with the split-half kernel (three MFMAs per product) far below those levels when the
    inputs are exactly representable in 8 bits.  Also: prologue / residual / statistics, bitwise repeatability and batch invariance,
    the C ABI's refusal of shapes the eight-wave kernel does not cover."""
    import torch
    import torch.nn.functional as F
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(91)
    B, H, W, C = 3, 32, 48, 128
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.03
    b = torch.randn(C, generator=g)
    sc_, sh_ = torch.rand(B, C, generator=g) + 0.5, torch.randn(B, C, generator=g) * 0.1
    res = torch.randn(B, H, W, C, generator=g)
    y = x.double() * sc_.double()[:, None, None, :] + sh_.double()[:, None, None, :]
    y = y * torch.sigmoid(y)
    ref = F.conv2d(y.permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    rms = float(ref.pow(2).mean().sqrt())
    xs, ws_, bs, scd, shd, resd = (t.cuda() for t in (x, w, b, sc_, sh_, res))
    outs = {}
    for name, code, gate in (('f16x2', ops.WSPLIT, 2e-5), ('fp16', ops.WF16, 4e-3), ('bf16', ops.WBF16, 3e-2)):
        pw = ops.pack_weight(ws_, bs, bf16=code)
        o = ops.conv2d(xs, pw, prologue=ops.PRO_AFFINE_SWISH, scale=scd, shift=shd, epilogue=ops.EPI_RESIDUAL, res=resd, emit_stats=True)
        o2 = ops.conv2d(xs, pw, prologue=ops.PRO_AFFINE_SWISH, scale=scd, shift=shd, epilogue=ops.EPI_RESIDUAL, res=resd, emit_stats=True)
        assert torch.equal(o, o2) and torch.equal(o._cf_stats.part, o2._cf_stats.part), name
        o1 = ops.conv2d(xs[1:2].contiguous(), pw, prologue=ops.PRO_AFFINE_SWISH, scale=scd[1:2].contiguous(), shift=shd[1:2].contiguous(),
                        epilogue=ops.EPI_RESIDUAL, res=resd[1:2].contiguous(), emit_stats=True)
        assert torch.equal(o[1:2], o1), name
        err = (o.double().cpu() - ref).abs()
        print(f'winograd {name}: max {float(err.max()):.3e} mean {float(err.mean()):.3e} (output rms {rms:.3f})')
        assert float(err.max()) <= gate * max(1.0, rms) * 4 and float(err.mean()) <= gate * max(1.0, rms) * 0.5, (name, float(err.max()), float(err.mean()))
        # statistics partials describe what was written (a thread adds its 16 values in fp32 before the fp64 partials)
        s = o._cf_stats
        tot = s.part.view(B, 32, s.parts, 2).sum(dim=2).cpu()
        grp = o.double().cpu().view(B, H * W, 32, C // 32)
        assert torch.allclose(tot[..., 0], grp.sum(dim=(1, 3)), rtol=1e-5, atol=5e-3) and \
            torch.allclose(tot[..., 1], grp.pow(2).sum(dim=(1, 3)), rtol=1e-5, atol=5e-3), name
        outs[name] = o
    # inputs exactly representable in 8 significant bits: V = B^T d B (sums of four such values) and U are not, but the single-operand
    # kernels must then sit within their rounding of U and V only -- and the data path (fragment order, scale, epilogue) is the split kernel's
    xq = (torch.randint(-8, 9, (1, 32, 32, 128), generator=g).float() / 8).cuda()
    wq = (torch.randint(-4, 5, (128, 128, 3, 3), generator=g).float() / 64).cuda()
    exact = F.conv2d(xq.double().permute(0, 3, 1, 2), wq.double(), padding=1).permute(0, 2, 3, 1)
    for code, tol in ((ops.WSPLIT, 1e-5), (ops.WF16, 1e-5), (ops.WBF16, 2e-2)):
        o = ops.conv2d(xq, ops.pack_weight(wq, None, bf16=code))
        # G g G^T of multiples of 1/64 has at most 8 significant bits (quarters of sums of nine small integers), B^T d B at most 7:
        # exact in IEEE half, so 'fp16' reproduces the convolution exactly here; bf16 (8 bits) rounds some U
        assert float((o.double() - exact).abs().max()) <= tol, (code, float((o.double() - exact).abs().max()))
    # shapes outside the eight-wave kernel: the host falls back to the direct 16-bit kernel, the C ABI refuses
    assert ops.conv_code(2, 128, 128, 32, 48) == ops.WF16 and ops.conv_code(1, 128, 128, 32, 48) == ops.WBF16
    assert ops.conv_code(2, 64, 64, 512, 512) == 2 and ops.conv_code(1, 512, 512, 16, 16) == 1 and ops.conv_code(2, 128, 128, 32, 32, up2x=True) == 2
    pw64 = ops.pack_weight(torch.randn(64, 64, 3, 3, device='cuda') * 0.05, None, bf16=ops.WF16)
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 32, 32, 64, device='cuda'), pw64)


def test_split_conv_stride2_against_fp64_and_the_exact_kernel(sc):
    """Downsample (vqgan_arch.py:117-126: zero row / column bottom / right, 3x3 stride 2) on split halves: the space-to-depth 2x2 form
    of cf_split.hip against an fp64 convolution and against the exact-fp32 kernel -- every encoder shape class (64-wide and 128-wide
    channel tiles, a 16x16 output), large un-normalised inputs through the range scale, statistics partials, batch invariance."""
    import torch
    import torch.nn.functional as F
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(1234)
    for (B, C, H, W, mag) in ((2, 64, 64, 64, 1.0), (3, 128, 32, 64, 300.0), (2, 256, 32, 32, 1e-3), (1, 16, 16, 32, 1.0), (2, 128, 128, 128, 5e4)):
        cout = max(C, 64) if C != 16 else 128
        x = (torch.randn(B, H, W, C, generator=g) * mag).cuda()
        w = (torch.randn(cout, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
        b = (torch.randn(cout, generator=g) * 0.1 * mag).cuda()
        ref = F.conv2d(F.pad(x.permute(0, 3, 1, 2).double(), (0, 1, 0, 1)), w.double(), b.double(), stride=2).permute(0, 2, 3, 1)
        pw = ops.pack_weight(w, b, bf16=ops.SPLIT, stride2=True)
        y = ops.conv2d(x, pw, stride=2, emit_stats=True, act=ops.act_scale(x))
        yf = ops.conv2d(x, ops.pack_weight(w, b), stride=2, emit_stats=True)
        rmax = float(ref.abs().max())
        es, ef = float((y.double() - ref).abs().max()), float((yf.double() - ref).abs().max())
        assert es <= 2e-5 * mag + 1e-5 * rmax and es <= 5.0 * ef + 1e-7 * rmax, (B, C, H, W, es, ef, rmax)
        # the epilogue's GroupNorm partials describe the tensor that was written
        if cout // 32 >= 2:
            sc_, sh_ = ops.groupnorm_tables([y], torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda'))
            yn = y.double().view(B, -1, 32, cout // 32)
            mean = yn.mean(dim=(1, 3))
            rstd = 1.0 / torch.sqrt(yn.var(dim=(1, 3), unbiased=False) + 1e-6)
            assert float((sc_.double().view(B, 32, -1)[:, :, 0] - rstd).abs().max() / rstd.abs().max()) <= 1e-5
            assert float((sh_.double().view(B, 32, -1)[:, :, 0] + mean * rstd).abs().max()) <= 1e-4 * max(1.0, float((mean * rstd).abs().max()))
        # bitwise: repeatable, and an image alone equals the image inside the batch
        y2 = ops.conv2d(x, pw, stride=2, emit_stats=True, act=ops.act_scale(x))
        assert torch.equal(y, y2) and torch.equal(y._cf_stats.part, y2._cf_stats.part)
        x1 = x[B - 1:B].contiguous()
        y1 = ops.conv2d(x1, pw, stride=2, emit_stats=True, act=ops.act_scale(x1))
        assert torch.equal(y[B - 1:B], y1)
    # refusals: the stride-2 form and the stride-2 descriptor belong together
    pw9 = ops.pack_weight(w, b, bf16=ops.SPLIT)
    with pytest.raises(ValueError):
        ops.conv2d(x, pw9, stride=2)
    with pytest.raises(ValueError):
        ops.conv2d(x, pw)


def test_split_conv_1x1_streaming_against_fp64_and_the_exact_kernel(sc):
    """ResBlock skip convolutions (vqgan_arch.py:150-164) on images: the 1x1 form of cf_split.hip (two activation slabs in flight)
    against fp64 and the exact-fp32 GEMM -- single and concatenated inputs, odd and even slab counts, both tile widths, un-normalised
    magnitudes through the range scale, residual epilogue, bitwise batch invariance; token-sized images keep the token GEMM."""
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(4321)
    for (B, c0, c1, cout, H, W, mag) in ((2, 128, 0, 64, 64, 64, 1.0), (2, 128, 128, 128, 64, 48, 400.0), (1, 96, 0, 128, 40, 64, 1e-3),
                                         (3, 32, 0, 64, 32, 64, 1.0), (2, 256, 256, 256, 32, 64, 3e4)):
        x = (torch.randn(B, H, W, c0, generator=g) * mag).cuda()
        x2 = (torch.randn(B, H, W, c1, generator=g) * mag * 0.1).cuda() if c1 else None
        cin = c0 + c1
        w = (torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5).cuda()
        b = (torch.randn(cout, generator=g) * 0.1 * mag).cuda()
        res = (torch.randn(B, H, W, cout, generator=g) * mag).cuda()
        xa = x if x2 is None else torch.cat((x, x2), dim=3)
        ref = xa.double() @ w.double().view(cout, cin).t() + b.double() + res.double()
        pw = ops.pack_weight(w, b, bf16=ops.SPLIT)
        assert pw.conv1 and pw.taps == 1
        act = ops.act_scale(xa.contiguous())
        y = ops.conv2d(x, pw, x2=x2, act=act, epilogue=ops.EPI_RESIDUAL, res=res)
        yf = ops.conv2d(x, ops.pack_weight(w, b), x2=x2, epilogue=ops.EPI_RESIDUAL, res=res)
        rmax = float(ref.abs().max())
        es, ef = float((y.double() - ref).abs().max()), float((yf.double() - ref).abs().max())
        assert es <= 2e-5 * mag + 1e-5 * rmax and es <= 5.0 * ef + 1e-7 * rmax, (c0, c1, cout, H, W, es, ef, rmax)
        assert torch.equal(y, ops.conv2d(x, pw, x2=x2, act=act, epilogue=ops.EPI_RESIDUAL, res=res))
        y1 = ops.conv2d(x[-1:].contiguous(), pw, x2=None if x2 is None else x2[-1:].contiguous(), act=act[-1:].contiguous(),
                        epilogue=ops.EPI_RESIDUAL, res=res[-1:].contiguous())
        assert torch.equal(y[-1:], y1)
    # a token-sized image with this weight form is refused on the host (the C ABI would read it as a token-GEMM weight)
    with pytest.raises(ValueError, match='1x1'):
        ops.conv2d(torch.zeros(1, 16, 16, 512, device='cuda'), pw)


def test_winograd_f43_forms_against_fp64(sc):
    """cf_wf43.hip (cf_conv_desc.winograd = 2, generator / fusion layers only): the 8-wave 64-channel form, the 16-wave 128-channel form on
    32-channel slabs and on 16-channel slabs (cin % 32 != 0), every prologue / epilogue / concat combination, extreme magnitudes behind
    the pack-time weight scale and the per-image activation scale -- against fp64: max error <= 2e-5 * max(|ref| / 4, 1) (measured
    0.8-1.5e-5: 5-8x the F(2,3) kernels, the conditioning of the larger transform), GroupNorm partials to 2e-6, bitwise repeatable;
    and the per-image bits do not depend on the batch."""
    import importlib.util
    import torch
    from codeformer_amd import ops
    spec = importlib.util.spec_from_file_location('f43_check', os.path.join(ROOT, 'tools', 'f43_check.py'))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    assert len(fc.SMALL) >= 12
    for c in fc.SMALL:
        assert fc.case(**c), c
    g = torch.Generator().manual_seed(3)
    for cin, cout in ((64, 64), (128, 128), (48, 128)):
        x = torch.randn(3, 32, 48, cin, generator=g).cuda()
        pw = ops.pack_weight((torch.randn(cout, cin, 3, 3, generator=g) * 0.03).cuda(), torch.randn(cout, generator=g).cuda(), bf16=ops.WF43)
        y = ops.conv2d(x, pw, emit_stats=True, act=ops.act_scale(x))
        x1 = x[1:2].contiguous()
        y1 = ops.conv2d(x1, pw, emit_stats=True, act=ops.act_scale(x1))
        assert torch.equal(y[1:2], y1) and torch.equal(y._cf_stats.part.view(3, -1)[1:2], y1._cf_stats.part.view(1, -1))


def test_process_level_ab_forms_are_bitwise_equal(sc):
    """Forms that a process-level switch selects (read once per process by the library): the 16-wave F(4,3) workgroup on 32-channel slabs
    (CF_F43_WIDE=k32, shipped) against 16-channel slabs (=k16) with IEEE-fp32 operands -- the same k groups in the same order on
    v_mfma_f32_16x16x4_f32, so the bits agree (split-half operands use another MFMA per slab width and do not) -- and the stride-2 form with /
    without its zero (tap, parity) blocks (CF_S2_SKIP=1 / 0): digests of outputs + GroupNorm partials from sub-processes must agree."""
    import re
    import subprocess
    import sys

    def digests(script, env, *args):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', script), *args], capture_output=True, text=True, env=e, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        d = re.findall(r'digest .*', r.stdout)
        assert len(d) >= 4 and all('False' not in x for x in d), r.stdout
        return d

    assert digests('f43_ovl_ab.py', {'CF_F43_WIDE': 'k32', 'F43_AB_DIGESTS_ONLY': '1'}, 'fp32') == \
        digests('f43_ovl_ab.py', {'CF_F43_WIDE': 'k16', 'F43_AB_DIGESTS_ONLY': '1'}, 'fp32')
    assert digests('s2_skip_ab.py', {'CF_S2_SKIP': '1'}) == digests('s2_skip_ab.py', {'CF_S2_SKIP': '0'})


def test_winograd_f43_fp32_operands_against_fp64(sc):
    """The same kernel with IEEE-fp32 operands (CF_OPERAND_F32 + winograd = 2, ABI v20: precision 'fp32' of the generator / fusion layers):
    the same cases and batch invariance, bound 4e-5 * max(|ref| / 4, 1) (measured 1.1-2.8e-5); no weight / activation scale is involved
    (act tables are ignored)."""
    import importlib.util
    import torch
    from codeformer_amd import ops
    spec = importlib.util.spec_from_file_location('f43_check', os.path.join(ROOT, 'tools', 'f43_check.py'))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    for c in fc.SMALL:
        assert fc.case(fp32=True, **c), c
    g = torch.Generator().manual_seed(4)
    for cin, cout in ((64, 64), (128, 128), (48, 128)):
        x = torch.randn(3, 32, 48, cin, generator=g).cuda()
        pw = ops.pack_weight((torch.randn(cout, cin, 3, 3, generator=g) * 0.03).cuda(), torch.randn(cout, generator=g).cuda(), bf16=ops.WF43F)
        assert pw.wino == 2 and not pw.bf16 and not ops.needs_act_scale(pw)
        y = ops.conv2d(x, pw, emit_stats=True)
        y1 = ops.conv2d(x[1:2].contiguous(), pw, emit_stats=True)
        assert torch.equal(y[1:2], y1) and torch.equal(y._cf_stats.part.view(3, -1)[1:2], y1._cf_stats.part.view(1, -1))


def test_winograd_f43_fp32_upsampling_gather(sc):
    """Round 6: nearest x2 + 3x3 (Upsample, vqgan_arch.py:129-138) on the fp32 F(4x4,3x3) kernel whose gather reads source pixel (y >> 1, x >> 1):
    against fp64 within the fp32-operand bound of the plain form (4e-5 * max(|ref| / 4, 1)), next to the folded sub-pixel kernel it replaces in
    precision 'fp32'; GroupNorm partials of the output; batch invariance; refusals."""
    import pytest
    import torch
    import torch.nn.functional as F
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(17)
    for (B, H, W, cin, cout) in ((2, 32, 32, 128, 128), (1, 64, 48, 256, 256), (3, 16, 32, 64, 128)):
        x = torch.randn(B, H, W, cin, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.conv2d(F.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2.0, mode='nearest'), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        pw = ops.pack_weight(w.cuda(), b.cuda(), bf16=ops.WF43F)
        y = ops.conv2d(x.cuda(), pw, upsample=True, emit_stats=True)
        folded = ops.conv2d(x.cuda(), ops.pack_weight(w.cuda(), b.cuda(), up2x=True), upsample=True, emit_stats=True)
        scale = float(ref.abs().max())
        err, err_f = float((y.cpu().double() - ref).abs().max()), float((folded.cpu().double() - ref).abs().max())
        print(f'F(4,3) upsampling gather {H}x{W} {cin}->{cout}: max err {err:.2e} (folded direct kernel {err_f:.2e}), |ref| max {scale:.2f}')
        assert tuple(y.shape) == (B, 2 * H, 2 * W, cout) and err <= 4e-5 * max(scale / 4.0, 1.0)
        st = y._cf_stats
        got = st.part.view(B, 32, st.parts, 2).sum(2).cpu()
        r = y.cpu().double().view(B, 4 * H * W, 32, st.cpg)
        want = torch.stack([r.sum((1, 3)), (r * r).sum((1, 3))], -1)
        room = torch.stack([r.abs().sum((1, 3)), (r * r).sum((1, 3))], -1)      # (a group's sum may cancel: measure against the sum of magnitudes)
        assert float(((got - want).abs() / room).max()) < 1e-6
        if B > 1:
            assert torch.equal(ops.conv2d(x[1:2].contiguous().cuda(), pw, upsample=True), y[1:2])
    assert ops.f43_up_ok(128, 128, 512, 512) and ops.f43_up_ok(256, 256, 64, 64) and not ops.f43_up_ok(512, 512, 32, 32) and not ops.f43_up_ok(128, 64, 512, 512)
    xs = torch.randn(1, 32, 32, 128, device='cuda')
    with pytest.raises((RuntimeError, ValueError)):      # split-half operands have no upsampling gather (their folded form is the faster one)
        ops.conv2d(xs, ops.pack_weight(torch.randn(128, 128, 3, 3, device='cuda') * 0.03, None, bf16=ops.WF43), upsample=True, act=ops.act_scale(xs))


def test_winograd_f43_with_512_input_channels(sc):
    """Round 6: the 16-wave form on 32-channel slabs keeps 512 GroupNorm rows in LDS (F4_TAB_32), so the 512 -> 256 fusion convolution at 64x64
    (concatenated [enc, dec], codeformer_arch.py:152) runs F(4x4,3x3) in precision 'fp32'.  Against fp64; the products of twice as many channels
    are summed, so the bound is sqrt(2) x the 4e-5 (fp32 operands) / 2e-5 (split halves) of the <= 256-channel cases; the 8-wave form still refuses."""
    import pytest
    import torch
    import torch.nn.functional as F
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(3)
    for (B, H, cin, cout, cs) in ((2, 64, 512, 256, 256), (1, 32, 512, 128, 256)):
        x = torch.randn(B, H, H, cin, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2 / (9 * cin)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        sc_, sh_ = torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g) * 0.1
        xd = x.double() * sc_.double()[:, None, None, :] + sh_.double()[:, None, None, :]
        ref = F.conv2d((xd * torch.sigmoid(xd)).permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        scale = max(float(ref.abs().max()) / 4.0, 1.0)
        for code, bound in ((ops.WF43F, 4e-5), (ops.WF43, 2e-5)):
            pw = ops.pack_weight(w.cuda(), b.cuda(), bf16=code)
            xc = x.cuda()
            y = ops.conv2d(xc[..., :cs].contiguous(), pw, x2=xc[..., cs:].contiguous(), prologue=ops.PRO_AFFINE_SWISH, scale=sc_.cuda(), shift=sh_.cuda(), emit_stats=True)
            err = float((y.cpu().double() - ref).abs().max())
            assert err <= 2 ** 0.5 * bound * scale, (code, cin, cout, err)
    assert ops.f43_ok(512, 256, 64, 64, fp32=True) and not ops.f43_ok(512, 64, 64, 64, fp32=True)
    pw64 = ops.pack_weight(torch.randn(64, 512, 3, 3, device='cuda') * 0.01, None, bf16=ops.WF43F)
    with pytest.raises(RuntimeError, match='input channels'):
        ops.conv2d(torch.randn(1, 64, 64, 512, device='cuda'), pw64)
