"""Micro-benchmark of cf_conv2d shapes/variants on the GPU (events on the launch stream).
usage: python tools/conv_bench.py [B]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codeformer_amd import ops  # noqa: E402
from codeformer_amd.ops import (EPI_NONE, EPI_RESIDUAL, PRO_AFFINE_SWISH, PRO_LEAKY, PRO_NONE, PRO_AFFINE)  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = 'cuda'


def bench(name, cin, cout, H, k=3, prologue=PRO_NONE, epilogue=EPI_NONE, upsample=False, stride=1, reps=5):
    x = torch.randn(B, H, H, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    pw = ops.pack_weight(w, torch.randn(cout, device=dev), up2x=upsample)
    sc = sh = res = None
    if prologue in (PRO_AFFINE, PRO_AFFINE_SWISH):
        sc, sh = torch.rand(B, cin, device=dev) + 0.5, torch.randn(B, cin, device=dev) * 0.1
    Ho = H * 2 if upsample else (H // 2 if stride == 2 else H)
    if epilogue == EPI_RESIDUAL:
        res = torch.randn(B, Ho, Ho, cout, device=dev)
    f = lambda: ops.conv2d(x, pw, prologue=prologue, scale=sc, shift=sh, epilogue=epilogue, res=res, upsample=upsample, stride=stride)
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * Ho * Ho * cout * cin * (4 if upsample else k * k)
    print(f'{name:52s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)


if len(sys.argv) > 2 and sys.argv[2] == 'brief':
    for cin, cout, H in ((128, 128, 256), (64, 64, 512)):
        bench(f'3x3 {cin}->{cout} @{H} plain', cin, cout, H)
        bench(f'3x3 {cin}->{cout} @{H} swish+res', cin, cout, H, prologue=PRO_AFFINE_SWISH, epilogue=EPI_RESIDUAL)
    sys.exit(0)
for cin, cout, H in ((64, 64, 512), (128, 128, 256), (256, 256, 64), (512, 512, 16), (256, 256, 32)):
    bench(f'3x3 {cin}->{cout} @{H} plain', cin, cout, H)
    bench(f'3x3 {cin}->{cout} @{H} swish', cin, cout, H, prologue=PRO_AFFINE_SWISH)
    bench(f'3x3 {cin}->{cout} @{H} swish+res', cin, cout, H, prologue=PRO_AFFINE_SWISH, epilogue=EPI_RESIDUAL)
    bench(f'3x3 {cin}->{cout} @{H} leaky', cin, cout, H, prologue=PRO_LEAKY)
bench('3x3 128->128 up @256->512', 128, 128, 256, upsample=True)
bench('1x1 512->512 @16', 512, 512, 16, k=1)
bench('1x1 512->1536 @16', 512, 1536, 16, k=1)
