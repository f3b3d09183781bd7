"""Per-stage cycle sums of one workgroup of the eight-wave Winograd kernel (library built with -DWS_TIMING=1, loaded through CF_LIB_PATH)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
B, cin, cout, H = 16, int(os.environ.get('CIN', 128)), 128, int(os.environ.get('H', 256))
x = torch.randn(B, H, H, cin, device='cuda')
pw = ops.pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, torch.randn(cout, device='cuda'), bf16=ops.WSPLIT)
sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
res = torch.randn(B, H, H, cout, device='cuda')
for _ in range(3):
    y = ops.conv2d(x, pw, prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=res, emit_stats=True)
torch.cuda.synchronize()
d = y.reshape(-1)[:64].cpu().view(8, 8)
print('wave: fill  mma  feed  barrier  epilogue  (shader cycles, sums over the %d slabs of one patch)' % int(d[0, 5]))
for w in range(8):
    f, m, fe, b, e = (int(v) for v in d[w, :5])
    print(f'  {w} (xi {w & 3}, {"MMA-first" if w < 4 else "feed-first"}): {f:7d} {m:7d} {fe:7d} {b:7d} {e:7d}   per slab: mma {m // int(d[0,5]):5d} feed {fe // int(d[0,5]):5d} barrier {b // int(d[0,5]):5d}')
