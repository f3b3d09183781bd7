"""Split-half conv kernels (cf_split.hip direct form, cf_winograd.hip / cf_wsplit.hip Winograd form; CF_OPERAND_F16X2) vs an fp64
reference, next to the exact-fp32 kernels (accuracy + time).
GPU box only.  usage: python tools/split_check.py [quick]"""
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


def case(B, H, W, cin, cout, *, upsample=False, c_split=None, prologue=ops.PRO_NONE, epilogue=ops.EPI_NONE, stats=False, seed=0,
         timing=True, wscale=1.0, xscale=1.0, others=True):
    """Returns (err_split, err_best_fp32_kernel, stats_rel_err, ref_absmax, mean_abs_err_split, err_wsplit, stats_rel_err_wsplit);
    the last two are NaN / 0 for shapes the Winograd form does not cover."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, cin, generator=g) * xscale
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5 * wscale
    b = torch.randn(cout, generator=g) * 0.1
    sc = torch.rand(B, cin, generator=g) + 0.5
    sh = torch.randn(B, cin, generator=g) * 0.1
    Ho, Wo = (2 * H, 2 * W) if upsample else (H, W)
    res = torch.randn(B, Ho, Wo, cout, generator=g)
    ss = torch.randn(B, Ho, Wo, cout, generator=g) * 0.3
    xd = x.double()
    if prologue in (ops.PRO_AFFINE, ops.PRO_AFFINE_SWISH):
        xd = xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]
        if prologue == ops.PRO_AFFINE_SWISH:
            xd = xd * torch.sigmoid(xd)
    elif prologue == ops.PRO_LEAKY:
        xd = F.leaky_relu(xd, 0.2)
    xn = xd.permute(0, 3, 1, 2)
    if upsample:
        xn = F.interpolate(xn, scale_factor=2.0, mode='nearest')
    ref = F.conv2d(xn, w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    if epilogue == ops.EPI_RESIDUAL:
        ref = ref + res.double()
    elif epilogue == ops.EPI_SFT:
        ref = res.double() + 0.7 * (res.double() * ss.double() + ref)
    kw = dict(prologue=prologue, epilogue=epilogue, emit_stats=stats, upsample=upsample)
    if prologue in (ops.PRO_AFFINE, ops.PRO_AFFINE_SWISH):
        kw.update(scale=sc.cuda(), shift=sh.cuda())
    if epilogue != ops.EPI_NONE:
        kw.update(res=res.cuda())
    if epilogue == ops.EPI_SFT:
        kw.update(sft_scale=ss.cuda(), sft_w=0.7)
    xc = x.cuda()
    x1, x2 = (xc, None) if c_split is None else (xc[..., :c_split].contiguous(), xc[..., c_split:].contiguous())
    pw_s = ops.pack_weight(w.cuda(), b.cuda(), bf16=ops.SPLIT, up2x=upsample)
    ys = ops.conv2d(x1, pw_s, x2=x2, **kw)
    d = (ys.cpu().double() - ref).abs()
    es, ems = float(d.max()), float(d.mean())
    msg = (f'B{B} {H}x{W} {cin}->{cout}{" up2x" if upsample else ""} pro{prologue} epi{epilogue}{" cat" if c_split else ""}: '
           f'split max {es:.2e} mean {ems:.2e}')
    pw_w = None
    ew, estw = float('nan'), 0.0
    if not upsample and ops.winograd_ok(cin, cout, H, W):   # Winograd form of the split-half scheme (cf_winograd.hip, H2)
        pw_w = ops.pack_weight(w.cuda(), b.cuda(), bf16=ops.WSPLIT)
        yw = ops.conv2d(x1, pw_w, x2=x2, **kw)
        dw = (yw.cpu().double() - ref).abs()
        ew = float(dw.max())
        msg += f' | wsplit max {ew:.2e} mean {float(dw.mean()):.2e}'
        if stats:
            sw = yw._cf_stats
            tw = sw.part.view(B, 32, sw.parts, 2).sum(2)
            r = yw.double().view(B, Ho * Wo, 32, sw.cpg)
            want = torch.stack([r.sum((1, 3)), (r * r).sum((1, 3))], -1)
            estw = float(((tw - want).abs() / want.abs().clamp_min(1e-6)).max())
            msg += f' (stats {estw:.1e})'
    ef = float('nan')
    pw_f = None
    if others:
        code = ops.conv_code(ops.WINOGRAD, cin, cout, H, W, up2x=upsample)
        pw_f = ops.pack_weight(w.cuda(), b.cuda(), bf16=code, up2x=upsample)
        yf = ops.conv2d(x1, pw_f, x2=x2, **kw)
        df = (yf.cpu().double() - ref).abs()
        ef = float(df.max())
        msg += f' | fp32 {"winograd" if code else "direct"} max {ef:.2e} mean {float(df.mean()):.2e}'
    msg += f' (ref max {float(ref.abs().max()):.2f}, weight scale 2^{pw_s.scale and __import__("math").frexp(pw_s.scale)[1] - 1})'
    est = 0.0
    if stats:   # the epilogue's GroupNorm partials must describe exactly the tensor that was written
        sw = ys._cf_stats
        tw = sw.part.view(B, 32, sw.parts, 2).sum(2)
        r = ys.double().view(B, Ho * Wo, 32, sw.cpg)
        want = torch.stack([r.sum((1, 3)), (r * r).sum((1, 3))], -1)
        est = float(((tw - want).abs() / want.abs().clamp_min(1e-6)).max())
        msg += f' | stats rel err {est:.1e}'
    if timing:
        ts_ = t_ms(lambda: ops.conv2d(x1, pw_s, x2=x2, **kw))
        fl = 2.0 * B * Ho * Wo * cout * cin * 9
        msg += f' | split {ts_:.3f} ms ({fl / ts_ / 1e9:.0f} TF-equiv)'
        if pw_w is not None:
            tw_ = t_ms(lambda: ops.conv2d(x1, pw_w, x2=x2, **kw))
            msg += f' wsplit {tw_:.3f} ms ({fl / tw_ / 1e9:.0f}) x{ts_ / tw_:.2f}'
        if pw_f is not None:
            tf_ = t_ms(lambda: ops.conv2d(x1, pw_f, x2=x2, **kw))
            msg += f' fp32 {tf_:.3f} ms ({fl / tf_ / 1e9:.0f}) x{tf_ / ts_:.2f}'
    print(msg, flush=True)
    return es, ef, est, float(ref.abs().max()), ems, ew, estw


CASES = [dict(B=1, H=16, W=16, cin=32, cout=64),
         dict(B=2, H=16, W=32, cin=64, cout=128, seed=1),
         dict(B=2, H=32, W=16, cin=64, cout=64, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=2),
         dict(B=2, H=32, W=32, cin=128, cout=64, c_split=64, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, seed=3),
         dict(B=2, H=16, W=16, cin=512, cout=512, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=4),
         dict(B=1, H=64, W=64, cin=256, cout=256, prologue=ops.PRO_AFFINE, stats=True, seed=5),
         dict(B=2, H=16, W=16, cin=128, cout=128, upsample=True, stats=True, seed=6),
         dict(B=1, H=32, W=16, cin=64, cout=64, upsample=True, seed=7),
         dict(B=1, H=32, W=32, cin=256, cout=128, c_split=128, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=8),
         dict(B=2, H=32, W=48, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=21),   # 8-wave Winograd form
         dict(B=1, H=32, W=32, cin=64, cout=256, c_split=32, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=True, seed=22),
         dict(B=1, H=40, W=32, cin=32, cout=128, seed=23),                                                                          # two slabs
         # magnitudes: large / tiny weights and activations (the pack-time power-of-two scale must absorb them)
         dict(B=1, H=16, W=16, cin=64, cout=64, wscale=300.0, seed=9),
         dict(B=1, H=16, W=16, cin=64, cout=64, wscale=1e-4, xscale=30.0, seed=10)]

if __name__ == '__main__':
    for c in CASES:
        case(timing=False, **c)
    if len(sys.argv) > 1 and sys.argv[1] == 'quick':
        sys.exit(0)
    for shape in ((16, 256, 256, 128, 128), (16, 512, 512, 64, 64), (16, 64, 64, 256, 256), (16, 128, 128, 128, 128),
                  (16, 32, 32, 256, 256), (16, 16, 16, 512, 512)):
        case(*shape, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9)
    case(16, 256, 256, 256, 128, c_split=128, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=11)
    case(16, 256, 256, 128, 128, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=True, seed=12)
    case(16, 256, 256, 128, 128, seed=13)
    case(16, 256, 256, 128, 128, upsample=True, stats=True, seed=14)
    case(16, 128, 128, 256, 256, upsample=True, stats=True, seed=15)
