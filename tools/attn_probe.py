"""cf_attention (8 heads x 64) per launch, timed inside a captured graph of 20 launches (no host launch floor); CF_LIB_PATH picks the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
for B in (1, 2, 4, 16):
    qk = torch.randn(B * 256, 1024, device='cuda')
    v = torch.randn(B * 256, 512, device='cuda')
    ref = None
    f = lambda: ops.attention(qk[:, :512], qk[:, 512:], v, B, 8, 64, 0.125)
    for _ in range(3):
        y = f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(20):
                y = f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    q, k = qk[:, :512].view(B, 256, 8, 64).permute(0, 2, 1, 3).double(), qk[:, 512:].view(B, 256, 8, 64).permute(0, 2, 1, 3).double()
    vv = v.view(B, 256, 8, 64).permute(0, 2, 1, 3).double()
    r = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ vv).permute(0, 2, 1, 3).reshape(B * 256, 512)
    print(f'B={B:2d}: {sorted(ts)[3] * 1e3:7.1f} us per launch   err vs fp64 {float((y.double() - r).abs().max()):.2e}', flush=True)
