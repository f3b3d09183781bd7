"""Downsample convolutions of the encoder: exact-fp32 kernel vs the stride-2 form of the split-half kernel (time per launch, error vs fp64)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
for B in (16, 1):
    for C, H in ((64, 512), (128, 256), (128, 128), (256, 64), (256, 32)):
        x = torch.randn(B, H, H, C, device='cuda')
        w = torch.randn(C, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
        b = torch.randn(C, device='cuda') * 0.1
        ref = F.conv2d(F.pad(x[:1].permute(0, 3, 1, 2).double(), (0, 1, 0, 1)), w.double(), b.double(), stride=2).permute(0, 2, 3, 1)
        res = []
        for name, pw, kw in (('fp32', ops.pack_weight(w, b), {}), ('f16x2', ops.pack_weight(w, b, bf16=ops.SPLIT, stride2=True), dict(act=ops.act_scale(x)))):
            for _ in range(3):
                y = ops.conv2d(x, pw, stride=2, emit_stats=True, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    y = ops.conv2d(x, pw, stride=2, emit_stats=True, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            res.append(f'{name} {sorted(ts)[2] * 1e3:7.1f} us err {float((y[:1].double() - ref).abs().max()):.1e}')
        print(f'B={B:2d} {C}ch {H}->{H // 2}: ' + '   '.join(res), flush=True)
