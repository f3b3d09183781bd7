"""1x1 skip convolutions of the channel-changing ResBlocks: fp32-MFMA GEMM vs the split-half token GEMM (cf_gemm_split.hip)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
B = int(os.environ.get('B', 16))
for cin, cout, H in ((128, 64, 512), (256, 128, 256), (256, 128, 128), (512, 256, 64), (512, 256, 32), (128, 256, 64)):
    x = torch.randn(B, H, H, cin, device='cuda')
    w = torch.randn(cout, cin, 1, 1, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    res = {}
    for name, code in (('fp32', 0), ('f16x2', ops.GSPLIT)):
        pw = ops.pack_weight(w, b, bf16=code)
        for _ in range(3):
            y = ops.conv2d(x, pw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = ops.conv2d(x, pw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        res[name] = (sorted(ts)[2], y)
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), b.double()).permute(0, 2, 3, 1)
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(f'{cin}->{cout} @{H}: fp32 {res["fp32"][0]:.3f} ms  f16x2 {res["f16x2"][0]:.3f} ms   ({gb:.2f} GB: {gb / res["f16x2"][0]:.0f} GB/s... TB/s={gb/res["f16x2"][0]:.2f})  '
          f'err fp32 {float((res["fp32"][1][:1].double() - ref).abs().max()):.2e} f16x2 {float((res["f16x2"][1][:1].double() - ref).abs().max()):.2e}', flush=True)
