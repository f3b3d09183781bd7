"""EXPERIMENT (round 6, recorded in profiles/r06_wsplit_vs_4wave.txt): is the four-wave split-half F(2,3) kernel (64-channel tiles, twice the
workgroups) bitwise the eight-wave one, and which is faster for one face?  Needs a library whose cf_wsplit_covers() returns false for
CF_OPERAND_F16X2 when CF_WSPLIT_OFF is set (three lines, not in the product); run twice: plain, and with CF_WSPLIT_OFF=1.
Result: the SAME BITS (outputs and GroupNorm partials), one face 35.5 vs 36.7 us (64x64) / 33.9 vs 34.9 us (32x32) per launch -- the chain
of K slabs sets the latency, not the tile width -- and sixteen faces 17-23 % slower: the host rule stays."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util  # noqa: E402

from codeformer_amd import ops  # noqa: E402

spec = importlib.util.spec_from_file_location('f43_check', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'f43_check.py'))
fc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fc)
print('CF_WSPLIT_OFF =', os.environ.get('CF_WSPLIT_OFF'))
for B in (1, 16):
    for (H, cin, cout, kw) in ((64, 256, 256, dict(prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL)), (32, 256, 256, dict(prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL)),
                               (64, 512, 256, dict(prologue=ops.PRO_AFFINE_SWISH, c_split=256)), (32, 256, 256, dict(prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT))):
        x1, x2, w, b, k, _ = fc.make(B, H, H, cin, cout, stats=True, seed=3, ref=False, **kw)
        pw = ops.pack_weight(w, b, bf16=ops.WSPLIT)
        y = ops.conv2d(x1, pw, x2=x2, **k)
        torch.cuda.synchronize()
        h = hashlib.sha256(y.cpu().numpy().tobytes())
        h.update(y._cf_stats.part.cpu().numpy().tobytes())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.conv2d(x1, pw, x2=x2, **k)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f'B={B} {H}x{H} {cin}->{cout} pro{k["prologue"]} epi{k["epilogue"]}: digest {h.hexdigest()[:16]}  {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per launch', flush=True)
