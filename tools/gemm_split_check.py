"""Split-half token GEMM (cf_gemm_split.hip) vs fp64 next to the exact fp32 GEMM: accuracy, bitwise independence of the split count,
time.  GPU box only.  usage: python tools/gemm_split_check.py [quick]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops  # noqa: E402


def t_ms(fn, n=20):
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


def case(M, K, N, epilogue=ops.EPI_NONE, bias=True, seed=0, timing=False, wscale=1.0):
    """Returns (err_split, err_fp32, ref_absmax, bits_equal_over_split_counts)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5 * wscale
    b = torch.randn(N, generator=g) * 0.1 if bias else None
    res = torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().t() + (0 if b is None else b.double())
    if epilogue == ops.EPI_GELU:
        ref = F.gelu(ref)
    elif epilogue == ops.EPI_RESIDUAL:
        ref = ref + res.double()
    kw = dict(epilogue=epilogue, res=res.cuda() if epilogue == ops.EPI_RESIDUAL else None)
    xc = x.cuda()
    pw_s = ops.pack_weight(w.cuda(), None if b is None else b.cuda(), bf16=ops.GSPLIT)
    pw_f = ops.pack_weight(w.cuda(), None if b is None else b.cuda())
    ys = ops.linear(xc, pw_s, **kw)
    yf = ops.linear(xc, pw_f, **kw)
    es, ef = float((ys.cpu().double() - ref).abs().max()), float((yf.cpu().double() - ref).abs().max())
    # the same bits for every split count the shape allows
    same = True
    x4 = xc.view(M // 256, 16, 16, K)
    r4 = None if kw['res'] is None else kw['res'].view(M // 256, 16, 16, N)
    for ns in (-1, 1, 2, 4, 8):    # (-1: the chunks shared by the waves of one workgroup, ABI v21)
        if ns == -1 and K > 1024:
            continue
        if ns == -1 or (K // 128) % ns == 0:
            y = ops.conv2d(x4, pw_s, epilogue=epilogue, res=r4, split_k=ns).view(M, N)
            same = same and bool(torch.equal(y, ys))
    msg = f'M{M} K{K} N{N} epi{epilogue}: split max {es:.2e} | fp32 max {ef:.2e} (ref max {float(ref.abs().max()):.2f}) | split counts bitwise equal: {same}'
    if timing:
        ts_, tf_ = t_ms(lambda: ops.linear(xc, pw_s, **kw)), t_ms(lambda: ops.linear(xc, pw_f, **kw))
        fl = 2.0 * M * N * K
        msg += f' | split {ts_ * 1e3:.1f} us ({fl / ts_ / 1e9:.0f} TF-equiv) fp32 {tf_ * 1e3:.1f} us ({fl / tf_ / 1e9:.0f}) x{tf_ / ts_:.2f}'
        per = {}
        for ns in (-1, 1, 2, 4):
            if (ns == -1 and K <= 1024) or (ns > 0 and (K // 128) % ns == 0):
                per[ns] = t_ms(lambda: ops.conv2d(x4, pw_s, epilogue=epilogue, res=r4, split_k=ns)) * 1e3
        msg += ' | us by split_k: ' + ' '.join(f'{k}:{v:.1f}' for k, v in per.items())
    print(msg, flush=True)
    return es, ef, float(ref.abs().max()), same


CASES = [dict(M=256, K=256, N=512), dict(M=512, K=512, N=1024, epilogue=ops.EPI_GELU, seed=1), dict(M=256, K=1024, N=512, epilogue=ops.EPI_RESIDUAL, seed=2),
         dict(M=768, K=512, N=1024, bias=False, seed=3), dict(M=256, K=512, N=64, seed=4, wscale=200.0), dict(M=256, K=128, N=128, seed=5, wscale=1e-3)]

if __name__ == '__main__':
    for c in CASES:
        case(**c)
    if len(sys.argv) > 1 and sys.argv[1] == 'quick':
        sys.exit(0)
    for M in (256, 1024, 4096):
        case(M, 512, 512, epilogue=ops.EPI_RESIDUAL, timing=True, seed=7)
        case(M, 512, 1024, epilogue=ops.EPI_GELU, timing=True, seed=8)
        case(M, 1024, 512, epilogue=ops.EPI_RESIDUAL, timing=True, seed=9)
        case(M, 256, 512, timing=True, seed=10)
