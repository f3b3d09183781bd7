"""Tile-size probe for the Transformer's token GEMMs (GPU box): the 64x64 split-K kernel the network uses at every batch size against the
128-wide-tile kernel (split_k=0) at 4096 / 1024 / 256 tokens.  Result (round 2): within +-4 % at 4096 tokens, 1.5-2.8x slower at 256.
usage: python tools/gemm_tile_probe.py"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from codeformer_amd import ops
torch.manual_seed(0)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (16, 4, 1):
    for K, N in ((512, 1024), (512, 512), (1024, 512), (512, 1536)):
        x = torch.randn(B, 16, 16, K, device='cuda')
        pw = ops.pack_weight(torch.randn(N, K, device='cuda') * 0.05, torch.randn(N, device='cuda'))
        a = t(lambda: ops.conv2d(x, pw))
        b = t(lambda: ops.conv2d(x, pw, split_k=0))
        fl = 2.0 * B * 256 * K * N
        print(f'B={B} {K}->{N}: split-K kernel (64x64) {a:.1f} us = {fl/a/1e6:.1f} TF | large-tile kernel {b:.1f} us = {fl/b/1e6:.1f} TF')
