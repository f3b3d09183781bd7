"""fp32 token GEMMs at sixteen faces (GPU box): the 128-token tile kernel of round 6 (gemm_f32_tile_kernel, what split_k = 1 runs) against the
64x64 split-K instantiation of cf_igemm.hip it replaces there (split_k = 2: the same bits, two workgroups per tile; and, with
CF_GEMM_F32_TILE=0 in the environment, split_k = 1 on the old kernel).  Checks bitwise equality, prints us per launch.
usage: python tools/gemm_f32_tile_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops  # noqa: E402

torch.manual_seed(0)


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('CF_GEMM_F32_TILE =', os.environ.get('CF_GEMM_F32_TILE', '(unset: tile kernel on)'))
for B in (16, 8):
    for K, N, epi in ((256, 512, ops.EPI_NONE), (512, 1024, ops.EPI_NONE), (512, 512, ops.EPI_RESIDUAL), (512, 1024, ops.EPI_GELU), (1024, 512, ops.EPI_RESIDUAL)):
        x = torch.randn(B, 16, 16, K, device='cuda')
        res = torch.randn(B, 16, 16, N, device='cuda') if epi == ops.EPI_RESIDUAL else None
        pw = ops.pack_weight(torch.randn(N, K, device='cuda') * 0.05, torch.randn(N, device='cuda'))
        y1 = ops.conv2d(x, pw, epilogue=epi, res=res, split_k=1)
        y2 = ops.conv2d(x, pw, epilogue=epi, res=res, split_k=2)
        same = bool(torch.equal(y1, y2))
        a = t(lambda: ops.conv2d(x, pw, epilogue=epi, res=res, split_k=1))
        b = t(lambda: ops.conv2d(x, pw, epilogue=epi, res=res, split_k=2))
        fl = 2.0 * B * 256 * K * N
        print(f'B={B} {K}->{N} epi{epi}: split_k=1 {a:.1f} us = {fl / a / 1e6:.1f} TF | split_k=2 (igemm SK) {b:.1f} us = {fl / b / 1e6:.1f} TF | bitwise equal {same}')
