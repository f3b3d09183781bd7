"""Token GEMM kernels side by side (cf_gemm_split.hip): the tiled kernel / cross-workgroup split (split_k >= 1) against the in-workgroup chunk kernel
(split_k = -1; CF_GEMM_CHUNK_TILE=1 / 2 forces its 32 x 32 / 64 x 64 tile), timed inside a captured graph of 20 launches (no host gaps), outputs compared
bitwise.  GPU box only.   usage: python tools/gemm_chunk_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops  # noqa: E402


def graph_us(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


gen = torch.Generator(device='cuda').manual_seed(0)
for M in (256, 512, 1024, 2048, 4096, 8192):
    for K, N, epi in ((512, 512, ops.EPI_RESIDUAL), (512, 1024, ops.EPI_GELU), (1024, 512, ops.EPI_RESIDUAL), (512, 1536, ops.EPI_NONE)):
        x = torch.randn(M, K, device='cuda', generator=gen)
        w = torch.randn(N, K, device='cuda', generator=gen) * K ** -0.5
        b = torch.randn(N, device='cuda', generator=gen) * 0.1
        res = torch.randn(M, N, device='cuda', generator=gen)
        pw = ops.pack_weight(w, b, bf16=ops.GSPLIT)
        x4 = x.view(M // 256, 16, 16, K)
        r4 = res.view(M // 256, 16, 16, N) if epi == ops.EPI_RESIDUAL else None
        out = {}
        ts = {}
        for ns in (1, ops.splitk_for(pw, 16, 16, K, M // 256) or 1, -1):
            if ns in ts:
                continue
            out[ns] = ops.conv2d(x4, pw, epilogue=epi, res=r4, split_k=ns).clone()
            ts[ns] = graph_us(lambda: ops.conv2d(x4, pw, epilogue=epi, res=r4, split_k=ns))
        same = all(torch.equal(v, out[1]) for v in out.values())
        print(f'M{M:5d} K{K:5d} N{N:5d}: ' + '  '.join(f'split_k {k:2d}: {v:6.1f} us' for k, v in ts.items()) + f'   bitwise equal {same}', flush=True)
