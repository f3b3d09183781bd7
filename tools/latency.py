"""Faces-per-call sweep (the reference calls the network one face at a time, inference_codeformer.py:197-205): ms per call and
faces/s for B in 1..32, eager launches vs HIP-graph replay, per precision mode.  usage: python tools/latency.py [modes] [profile]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import codeformer_amd.archs  # noqa: E402,F401
from codeformer_amd import ops  # noqa: E402
from codeformer_amd.utils.registry import ARCH_REGISTRY  # noqa: E402
from oracle.synth import seeded_input  # noqa: E402

modes = (sys.argv[1] if len(sys.argv) > 1 else 'f16x2,fp32').split(',')
torch.manual_seed(0)
net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=['32', '64', '128', '256']).eval().cuda()
xs = seeded_input(32).cuda()


def bench(x, n):
    for _ in range(3):
        net(x, w=0.5, adain=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net(x, w=0.5, adain=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for mode in modes:
    net.precision = mode
    for graphs in (False, True, 'auto'):
        net.use_hip_graphs = graphs
        row = []
        for B in (1, 2, 4, 8, 16, 32):
            dt = bench(xs[:B].contiguous(), 10 if B <= 4 else 5)
            row.append(f'B={B}: {dt * 1e3:6.2f} ms {B / dt:6.1f}/s')
        print(f'{mode:6s} {"auto " if graphs == "auto" else "graph" if graphs else "eager"} | ' + ' | '.join(row), flush=True)
net.use_hip_graphs = False
if len(sys.argv) > 2:      # per-launch timing of one B=1 forward: where the time goes
    net.precision = modes[0]
    x = xs[:1].contiguous()
    for _ in range(2):
        net(x, w=0.5, adain=True)
    torch.cuda.synchronize()
    ops.PROFILE = []
    net(x, w=0.5, adain=True)
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for kind, fl, nb, e0, e1, shape in rec:
        a = agg.setdefault((kind,) + tuple(shape), [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = sum(v[1] for v in agg.values())
    print(f'B=1 conv/gemm launches: {sum(v[0] for v in agg.values())}, {tot:.2f} ms inside them')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f'   {str(k):60s} x{v[0]:2d} {v[1]:7.3f} ms')
