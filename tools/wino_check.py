"""Winograd conv kernel vs fp64 reference and vs the direct kernel (accuracy + time).  GPU box only."""
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


def case(B, H, W, cin, cout, *, c_split=None, prologue=ops.PRO_NONE, epilogue=ops.EPI_NONE, stats=False, seed=0, timing=True,
         direct=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    sc = torch.rand(B, cin, generator=g) + 0.5
    sh = torch.randn(B, cin, generator=g) * 0.1
    res = torch.randn(B, H, W, cout, generator=g)
    ss = torch.randn(B, H, W, cout, generator=g) * 0.3
    xd = x.double()
    if prologue in (ops.PRO_AFFINE, ops.PRO_AFFINE_SWISH):
        xd = xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]
        if prologue == ops.PRO_AFFINE_SWISH:
            xd = xd * torch.sigmoid(xd)
    elif prologue == ops.PRO_LEAKY:
        xd = F.leaky_relu(xd, 0.2)
    ref = F.conv2d(xd.permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    if epilogue == ops.EPI_RESIDUAL:
        ref = ref + res.double()
    elif epilogue == ops.EPI_SFT:
        ref = res.double() + 0.7 * (res.double() * ss.double() + ref)
    kw = dict(prologue=prologue, epilogue=epilogue, emit_stats=stats)
    if prologue in (ops.PRO_AFFINE, ops.PRO_AFFINE_SWISH):
        kw.update(scale=sc.cuda(), shift=sh.cuda())
    if epilogue != ops.EPI_NONE:
        kw.update(res=res.cuda())
    if epilogue == ops.EPI_SFT:
        kw.update(sft_scale=ss.cuda(), sft_w=0.7)
    xc = x.cuda()
    x1, x2 = (xc, None) if c_split is None else (xc[..., :c_split].contiguous(), xc[..., c_split:].contiguous())
    pw_d = ops.pack_weight(w.cuda(), b.cuda())
    pw_w = ops.pack_weight(w.cuda(), b.cuda(), bf16=ops.WINOGRAD)
    yw = ops.conv2d(x1, pw_w, x2=x2, **kw)
    yd = ops.conv2d(x1, pw_d, x2=x2, **kw) if direct else yw   # (direct=False: sizes / options the direct kernel does not combine)
    ed = float((yd.cpu().double() - ref).abs().max())
    ew = float((yw.cpu().double() - ref).abs().max())
    msg = f'B{B} {H}x{W} {cin}->{cout} pro{prologue} epi{epilogue}{" cat" if c_split else ""}: direct {ed:.2e} winograd {ew:.2e} (ref max {float(ref.abs().max()):.2f})'
    es = 0.0
    if stats:   # the epilogue's GroupNorm partials must describe exactly the tensor that was written
        sw = yw._cf_stats
        tw = sw.part.view(B, 32, sw.parts, 2).sum(2)
        r = yw.double().view(B, H * W, 32, sw.cpg)
        want = torch.stack([r.sum((1, 3)), (r * r).sum((1, 3))], -1)
        es = float(((tw - want).abs() / want.abs().clamp_min(1e-6)).max())
        msg += f' | stats rel err {es:.1e}'
    if timing:
        td_, tw_ = t_ms(lambda: ops.conv2d(x1, pw_d, x2=x2, **kw)), t_ms(lambda: ops.conv2d(x1, pw_w, x2=x2, **kw))
        fl = 2.0 * B * H * W * cout * cin * 9
        msg += f' | {td_:.3f} ms ({fl / td_ / 1e9:.0f} TF) -> {tw_:.3f} ms ({fl / tw_ / 1e9:.0f} TF-equiv) x{td_ / tw_:.2f}'
    print(msg, flush=True)
    return ed, ew, es, float(ref.abs().max())


if __name__ == '__main__':
    case(1, 16, 16, 16, 64, timing=False)
    case(2, 16, 32, 32, 64, timing=False, seed=1)
    case(2, 16, 16, 64, 128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, timing=False, seed=2)
    case(2, 32, 32, 128, 64, c_split=64, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=False, timing=False, seed=3)
    case(2, 16, 16, 512, 512, prologue=ops.PRO_AFFINE_SWISH, stats=True, timing=False, seed=4)
    case(1, 64, 64, 256, 256, prologue=ops.PRO_AFFINE, stats=True, timing=False, seed=5)
    for shape in ((16, 256, 256, 128, 128), (16, 512, 512, 64, 64), (16, 64, 64, 256, 256), (16, 128, 128, 128, 128),
                  (16, 16, 16, 512, 512), (16, 32, 32, 256, 256)):
        case(*shape, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9)
