"""Winograd F(4x4,3x3) split-half kernel (cf_wf43.hip, cf_conv_desc.winograd = 2) against an fp64 reference, next to the F(2x2,3x3)
split-half kernels (accuracy, GroupNorm partials, time).  GPU box only.
usage: python tools/f43_check.py [check] [time] [big] [fp32]      (default: check time; fp32: the IEEE-fp32-operand forms against the
fp32 F(2x2,3x3) kernel instead of the split-half ones)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops  # noqa: E402


def t_ms(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def make(B, H, W, cin, cout, c_split=None, prologue=ops.PRO_NONE, epilogue=ops.EPI_NONE, stats=False, seed=0, wscale=1.0, xscale=1.0,
         ref=True):
    dev = 'cpu' if ref else 'cuda'   # (timing-only cases draw their data on the device: no fp64 reference needed)
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, H, W, cin, generator=g, device=dev) * xscale
    w = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5 * wscale
    b = torch.randn(cout, generator=g, device=dev) * 0.1
    sc = torch.rand(B, cin, generator=g, device=dev) + 0.5
    sh = torch.randn(B, cin, generator=g, device=dev) * 0.1
    res = torch.randn(B, H, W, cout, generator=g, device=dev)
    ss = torch.randn(B, H, W, cout, generator=g, device=dev) * 0.3
    want = None
    if ref:
        xd = x.double()
        if prologue in (ops.PRO_AFFINE, ops.PRO_AFFINE_SWISH):
            xd = xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]
            if prologue == ops.PRO_AFFINE_SWISH:
                xd = xd * torch.sigmoid(xd)
        elif prologue == ops.PRO_LEAKY:
            xd = F.leaky_relu(xd, 0.2)
        want = F.conv2d(xd.permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        if epilogue == ops.EPI_RESIDUAL:
            want = want + res.double()
        elif epilogue == ops.EPI_SFT:
            want = res.double() + 0.7 * (res.double() * ss.double() + want)
    kw = dict(prologue=prologue, epilogue=epilogue, emit_stats=stats)
    if prologue in (ops.PRO_AFFINE, ops.PRO_AFFINE_SWISH):
        kw.update(scale=sc.cuda(), shift=sh.cuda())
    if epilogue != ops.EPI_NONE:
        kw.update(res=res.cuda())
    if epilogue == ops.EPI_SFT:
        kw.update(sft_scale=ss.cuda(), sft_w=0.7)
    xc = x.cuda()
    x1, x2 = (xc, None) if c_split is None else (xc[..., :c_split].contiguous(), xc[..., c_split:].contiguous())
    if prologue in (ops.PRO_NONE, ops.PRO_LEAKY):   # un-normalised input: the per-image range scale, as the arch modules pass it
        kw['act'] = ops.act_scale(x1, x2)
    return x1, x2, w.cuda(), b.cuda(), kw, want


def stats_err(y, B, H, W):
    st = y._cf_stats
    got = st.part.view(B, 32, st.parts, 2).sum(2)
    r = y.double().view(B, H * W, 32, st.cpg)
    want = torch.stack([r.sum((1, 3)), (r * r).sum((1, 3))], -1)
    return float(((got - want).abs() / want.abs().clamp_min(1e-6)).max())


def case(B, H, W, cin, cout, timing=False, check=True, fp32=False, **kwargs):
    x1, x2, w, b, kw, want = make(B, H, W, cin, cout, ref=check, **kwargs)
    pw4 = ops.pack_weight(w, b, bf16=ops.WF43F if fp32 else ops.WF43)
    pw2 = ops.pack_weight(w, b, bf16=ops.WINOGRAD if fp32 else ops.WSPLIT)
    msg = f'{"fp32 " if fp32 else ""}B{B} {H}x{W} {cin}->{cout} pro{kw["prologue"]} epi{kw["epilogue"]}{" cat" if x2 is not None else ""}:'
    ok = True
    if check:
        y4 = ops.conv2d(x1, pw4, x2=x2, **kw)
        y2 = ops.conv2d(x1, pw2, x2=x2, **kw)
        d4, d2 = (y4.cpu().double() - want).abs(), (y2.cpu().double() - want).abs()
        scale = float(want.abs().max())
        msg += f' F(4,3) max {float(d4.max()):.2e} mean {float(d4.mean()):.2e} | F(2,3) max {float(d2.max()):.2e} mean {float(d2.mean()):.2e} (ref max {scale:.3g})'
        # (fp32 operands: measured 1.1-2.8e-5 -- products rounded to 24 bits where the split-half sum of three carries ~32 -- bound 4e-5)
        ok = float(d4.max()) <= (4e-5 if fp32 else 2e-5) * max(scale / 4.0, 1.0) and bool(torch.isfinite(y4).all())
        if kw['emit_stats']:
            e4 = stats_err(y4, B, H, W)
            msg += f' stats {e4:.1e}'
            ok = ok and e4 < 2e-6   # (fp32 over four values, then fp64: the shipped kernels' scheme)
        y4b = ops.conv2d(x1, pw4, x2=x2, **kw)
        if not torch.equal(y4, y4b):
            msg += ' NOT REPRODUCIBLE'
            ok = False
    if timing:
        t4 = t_ms(lambda: ops.conv2d(x1, pw4, x2=x2, **kw))
        t2 = t_ms(lambda: ops.conv2d(x1, pw2, x2=x2, **kw))
        fl = 2.0 * B * H * W * cout * cin * 9
        nb = 4.0 * B * H * W * (cin + cout * (1 + (kw['epilogue'] != ops.EPI_NONE) + (kw['epilogue'] == ops.EPI_SFT)))
        msg += f' | F(4,3) {t4:.3f} ms ({fl / t4 / 1e9:.0f} TF-equiv, {nb / t4 / 1e9:.2f} TB/s) F(2,3) {t2:.3f} ms  x{t2 / t4:.2f}'
    print(('ok   ' if ok else 'FAIL ') + msg, flush=True)
    return ok


SMALL = [dict(B=1, H=16, W=32, cin=16, cout=64),
         dict(B=2, H=32, W=32, cin=64, cout=64, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=2),
         dict(B=2, H=32, W=64, cin=128, cout=64, c_split=64, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=True, seed=3),
         dict(B=1, H=48, W=32, cin=32, cout=128, prologue=ops.PRO_AFFINE, stats=True, seed=4),
         dict(B=1, H=32, W=32, cin=256, cout=256, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=5),
         dict(B=3, H=16, W=64, cin=64, cout=64, c_split=48, seed=6),
         dict(B=1, H=32, W=32, cin=256, cout=64, prologue=ops.PRO_AFFINE_SWISH, seed=7),
         dict(B=2, H=32, W=48, cin=48, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=8),   # 16 waves on 16-channel slabs (cin % 32 != 0)
         dict(B=1, H=48, W=32, cin=128, cout=256, c_split=64, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=True, seed=13),  # 16 waves, 32-channel slabs, two channel tiles, concat
         # magnitudes: the pack-time weight scale and the per-image activation scale must absorb them
         dict(B=1, H=16, W=32, cin=64, cout=64, wscale=300.0, seed=9),
         dict(B=2, H=16, W=32, cin=64, cout=64, wscale=1e-4, xscale=1e9, seed=10),
         dict(B=1, H=16, W=32, cin=64, cout=64, xscale=3e-9, prologue=ops.PRO_LEAKY, seed=11)]
TIMED = [dict(B=16, H=512, W=512, cin=64, cout=64, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=16, H=512, W=512, cin=64, cout=64, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=9),
         dict(B=16, H=512, W=512, cin=128, cout=64, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=9),
         dict(B=16, H=256, W=256, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=16, H=256, W=256, cin=128, cout=128, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=True, seed=12),
         dict(B=16, H=128, W=128, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=16, H=64, W=64, cin=256, cout=256, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=16, H=32, W=32, cin=256, cout=256, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=1, H=512, W=512, cin=64, cout=64, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9)]

if __name__ == '__main__':
    modes = sys.argv[1:] or ['check', 'time']
    if modes == ['fp32']:
        modes += ['check', 'time']
    good, f32 = True, 'fp32' in modes
    if 'check' in modes:
        for c in SMALL:
            good &= case(fp32=f32, **c)
    if 'big' in modes:      # accuracy at full size (fp64 reference on the host: slow)
        for c in TIMED[:1] + TIMED[3:4]:
            good &= case(fp32=f32, **dict(c, B=2))
    if 'time' in modes:
        for c in TIMED:
            case(timing=True, check=False, fp32=f32, **c)
    sys.exit(0 if good else 1)
