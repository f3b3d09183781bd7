"""Launch one conv shape a few times (for rocprofv3 --pmc runs).  usage: conv_single.py cin cout H [B] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
cin, cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
x = torch.randn(B, H, H, cin, device='cuda')
pw = ops.pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, torch.randn(cout, device='cuda'))
sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
for _ in range(reps):
    ops.conv2d(x, pw, prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, emit_stats=True)
torch.cuda.synchronize()
