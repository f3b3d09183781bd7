"""A/B of the forms of the 16-wave F(4x4,3x3) workgroup (CF_F43_WIDE = k32 | k16, read once per process by cf_wf43.hip; `ovl`, the overlapped
form of round 5, needs a library built with tools/experiments/ablation_and_timing_macros.patch): digests of outputs + GroupNorm partials on a
few wide-layer shapes (k16 and ovl share arithmetic and summation order: equal digests; with fp32 operands k32 agrees too)
and launch times.  GPU box only.    usage: CF_F43_WIDE=<mode> python tools/f43_ovl_ab.py [fp32]"""
import hashlib
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codeformer_amd import ops  # noqa: E402

spec = importlib.util.spec_from_file_location('f43_check', os.path.join(ROOT, 'tools', 'f43_check.py'))
fc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fc)
fp32 = 'fp32' in sys.argv[1:]
mode = os.environ.get('CF_F43_WIDE', 'default')
CASES = [dict(B=2, H=32, W=48, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=8),
         dict(B=1, H=48, W=32, cin=128, cout=256, c_split=64, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=True, seed=13),
         dict(B=3, H=64, W=64, cin=64, cout=128, prologue=ops.PRO_AFFINE, stats=True, seed=4),
         dict(B=2, H=128, W=128, cin=256, cout=128, prologue=ops.PRO_NONE, stats=True, seed=5),
         dict(B=1, H=16, W=16, cin=32, cout=128, seed=6),
         dict(B=1, H=32, W=32, cin=48, cout=128, prologue=ops.PRO_AFFINE_SWISH, seed=7)]
for c in CASES:
    x1, x2, w, b, kw, _ = fc.make(ref=False, **c)
    pw = ops.pack_weight(w, b, bf16=ops.WF43F if fp32 else ops.WF43)
    y = ops.conv2d(x1, pw, x2=x2, **kw)
    torch.cuda.synchronize()
    h = hashlib.sha256(y.cpu().numpy().tobytes())
    if kw['emit_stats']:
        h.update(y._cf_stats.part.cpu().numpy().tobytes())
    print(f'digest {c["H"]}x{c["W"]} {c["cin"]}->{c["cout"]}: {h.hexdigest()[:16]} finite {bool(torch.isfinite(y).all())}', flush=True)
if os.environ.get('F43_AB_DIGESTS_ONLY'):
    sys.exit(0)
TIMED = [dict(B=16, H=256, W=256, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=16, H=256, W=256, cin=128, cout=128, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, stats=True, seed=12),
         dict(B=16, H=256, W=256, cin=256, cout=128, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=9),
         dict(B=16, H=128, W=128, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=16, H=128, W=128, cin=256, cout=256, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=16, H=64, W=64, cin=256, cout=256, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=1, H=256, W=256, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9),
         dict(B=1, H=128, W=128, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9)]
for c in TIMED:
    x1, x2, w, b, kw, _ = fc.make(ref=False, **c)
    pw = ops.pack_weight(w, b, bf16=ops.WF43F if fp32 else ops.WF43)
    t = fc.t_ms(lambda: ops.conv2d(x1, pw, x2=x2, **kw), n=20)
    print(f'time [{mode}{" fp32" if fp32 else ""}] B{c["B"]} {c["H"]}x{c["W"]} {c["cin"]}->{c["cout"]} pro{kw["prologue"]} epi{kw["epilogue"]}: {t:.4f} ms', flush=True)
