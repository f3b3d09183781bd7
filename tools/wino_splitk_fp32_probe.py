"""fp32 Winograd F(2,3) on the 16x16 latents at sixteen / eight faces: one, two or four workgroups per tile (split_k; the same bits) -- which is
fastest with fp32 operands?  (The host rule `ops.splitk_for` was measured with split-half operands in round 3.)  usage: python tools/wino_splitk_fp32_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops  # noqa: E402

torch.manual_seed(0)


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for code, name in ((ops.WINOGRAD, 'fp32'), (ops.WSPLIT, 'f16x2')):
    for B in (16, 8):
        for cin, cout in ((512, 512), (256, 512)):
            x = torch.randn(B, 16, 16, cin, device='cuda')
            w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.02
            pw = ops.pack_weight(w, torch.randn(cout, device='cuda'), bf16=code)
            sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
            res = torch.randn(B, 16, 16, cout, device='cuda')
            kw = dict(prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=res, emit_stats=True)
            ys, row = [], []
            for ns in (1, 2, 4):
                if (cin // 128) % ns:
                    continue
                ys.append(ops.conv2d(x, pw, split_k=ns, **kw))
                row.append(f'split {ns}: {t(lambda: ops.conv2d(x, pw, split_k=ns, **kw)):.1f} us')
            same = all(torch.equal(y, ys[0]) for y in ys[1:])
            print(f'{name} B={B} {cin}->{cout} @16x16: ' + ' | '.join(row) + f' | host picks {ops.splitk_for(pw, 16, 16, cin, B)} | same bits {same}', flush=True)
