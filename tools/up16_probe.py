"""The 16x16 -> 32x32 Upsample convolution (512 channels): exact-fp32 folded kernel vs the split-half folded kernel."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
for B in (1, 2, 16):
    C, H = 512, 16
    x = torch.randn(B, H, H, C, device='cuda')
    w = torch.randn(C, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, device='cuda') * 0.1
    ref = F.conv2d(F.interpolate(x[:1].permute(0, 3, 1, 2).double(), scale_factor=2.0, mode='nearest'), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    out = []
    for name, pw, kw in (('fp32', ops.pack_weight(w, b, up2x=True), {}), ('f16x2', ops.pack_weight(w, b, bf16=ops.SPLIT, up2x=True), dict(act=ops.act_scale(x)))):
        for _ in range(3):
            y = ops.conv2d(x, pw, upsample=True, emit_stats=True, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = ops.conv2d(x, pw, upsample=True, emit_stats=True, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        out.append(f'{name} {sorted(ts)[2] * 1e3:7.1f} us err {float((y[:1].double() - ref).abs().max()):.1e}')
    print(f'B={B:2d} {C}ch {H}->{2 * H}: ' + '   '.join(out), flush=True)
