"""One split-half conv launch sequence for profilers: python tools/split_one.py cin cout H swish up [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
cin, cout, H, swish, up = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
code = {'split': ops.SPLIT, 'wino': ops.WINOGRAD, 'wsplit': ops.WSPLIT, 'wf43': ops.WF43, 'direct': 0}[os.environ.get('CONV_KIND', 'split')]
B = 16
x = torch.randn(B, H, H, cin, device='cuda')
pw = ops.pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, torch.randn(cout, device='cuda'), bf16=code, up2x=bool(up))
sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
Ho = 2 * H if up else H
res = torch.randn(B, Ho, Ho, cout, device='cuda')
kw = dict(prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=res) if swish else {}
for _ in range(reps):
    ops.conv2d(x, pw, upsample=bool(up), emit_stats=True, **kw)
torch.cuda.synchronize()
