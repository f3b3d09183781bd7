"""1x1 skip convolutions of the channel-changing ResBlocks: the fp32-MFMA GEMM vs the streaming 1x1 form of the split-half convolution
kernel (cf_split.hip TAPS = 1).  Time per launch, achieved bytes / s over the algorithmic traffic, error vs fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
B = int(os.environ.get('B', 16))
for c0, c1, cout, H in ((128, 0, 64, 512), (128, 128, 128, 256), (128, 128, 128, 128), (256, 256, 256, 64), (64, 0, 128, 256), (128, 0, 256, 64)):
    cin = c0 + c1
    x = torch.randn(B, H, H, c0, device='cuda')
    x2 = torch.randn(B, H, H, c1, device='cuda') if c1 else None
    w = torch.randn(cout, cin, 1, 1, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    xa = x if x2 is None else torch.cat((x, x2), 3)
    act = ops.act_scale(xa)
    res = {}
    for name, pw, kw in (('fp32', ops.pack_weight(w, b), {}), ('f16x2', ops.pack_weight(w, b, bf16=ops.SPLIT), dict(act=act))):
        for _ in range(3):
            y = ops.conv2d(x, pw, x2=x2, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = ops.conv2d(x, pw, x2=x2, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        res[name] = (sorted(ts)[2], y)
    ref = xa[:1].double() @ w.double().view(cout, cin).t() + b.double()
    gb = (xa.numel() + y.numel()) * 4 / 1e9
    print(f'B={B} {c0}+{c1}->{cout} @{H}: fp32 {res["fp32"][0]:.3f} ms ({gb / res["fp32"][0]:.2f} TB/s)  f16x2 {res["f16x2"][0]:.3f} ms ({gb / res["f16x2"][0]:.2f} TB/s)  '
          f'err fp32 {float((res["fp32"][1][:1].double() - ref).abs().max()):.2e} f16x2 {float((res["f16x2"][1][:1].double() - ref).abs().max()):.2e}', flush=True)
