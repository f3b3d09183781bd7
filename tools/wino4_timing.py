"""Per-stage cycle sums of one workgroup of the four-wave Winograd split-half kernel (library built with -DCF_WTIMING=1, CF_LIB_PATH)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
B, cin, cout, H = 16, int(os.environ.get('CIN', 64)), 64, int(os.environ.get('H', 512))
x = torch.randn(B, H, H, cin, device='cuda')
pw = ops.pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, torch.randn(cout, device='cuda'), bf16=ops.WSPLIT)
sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
res = torch.randn(B, H, H, cout, device='cuda')
for _ in range(3):
    y = ops.conv2d(x, pw, prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=res, emit_stats=True)
torch.cuda.synchronize()
d = y.reshape(-1)[:32].cpu().view(4, 8)
n = int(d[0, 6])
print(f'wave: gather-store  barrier1  transform(+loads)  barrier2  MMA  epilogue   (shader cycles summed over the {n} slabs of one patch; per slab in brackets)')
for w in range(4):
    v = [int(t) for t in d[w, :6]]
    print(f'  {w}: ' + '  '.join(f'{t:6d} [{t // n:5d}]' for t in v[:5]) + f'  {v[5]:6d}')
