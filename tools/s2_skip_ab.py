"""Stride-2 form of the split-half kernel: digests of outputs + GroupNorm partials on a few shapes (incl. one whose slabs mix parities: no skipping
there).  CF_S2_SKIP=1 / 0 (read once per process by the library) must give the same digests.  GPU box only."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
g = torch.Generator(device='cuda').manual_seed(3)
for C, H in ((64, 64), (128, 32), (256, 32), (48, 32)):
    x = torch.randn(2, H, H, C, device='cuda', generator=g)
    w = torch.randn(C if C != 48 else 64, C, 3, 3, device='cuda', generator=g) * 0.05
    b = torch.randn(w.shape[0], device='cuda', generator=g)
    pw = ops.pack_weight(w, b, bf16=ops.SPLIT, stride2=True)
    y = ops.conv2d(x, pw, stride=2, emit_stats=True, act=ops.act_scale(x))
    h = hashlib.sha256(y.cpu().numpy().tobytes() + y._cf_stats.part.cpu().numpy().tobytes()).hexdigest()[:16]
    print('digest', C, H, h)
