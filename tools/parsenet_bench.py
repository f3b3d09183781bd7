"""Time ParseNet (the reference's face-parsing model: in/out 512, 19 classes) on the HIP path.  GPU box only.
    python tools/parsenet_bench.py [batch]       # 512x512 faces, default 16
Prints ms per batch for forward() (mask + image heads) and for parse_labels() (device argmax), with executed TFLOP/s
from ops.PROFILE events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops  # noqa: E402
from codeformer_amd.facelib.parsing.parsenet import ParseNet  # noqa: E402


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.manual_seed(0)
    net = ParseNet(in_size=512, out_size=512, parsing_ch=19).eval().cuda()
    x = torch.rand(b, 3, 512, 512, device='cuda') * 2 - 1
    t_fwd = timed(lambda: net(x))
    t_lab = timed(lambda: net.parse_labels(x))
    ops.PROFILE = []
    net.parse_labels(x)
    torch.cuda.synchronize()
    flops = sum(r[1] for r in ops.PROFILE)
    tconv = sum(r[3].elapsed_time(r[4]) for r in ops.PROFILE)
    n = len(ops.PROFILE)
    ops.PROFILE = None
    print(f'ParseNet 512x512 x{b}: forward {t_fwd:.2f} ms ({b / t_fwd * 1e3:.0f} faces/s), parse_labels {t_lab:.2f} ms '
          f'({b / t_lab * 1e3:.0f} faces/s); {n} conv launches, {flops / b / 1e9:.1f} executed GFLOP/face, '
          f'{flops / tconv / 1e9:.1f} TFLOP/s inside the convs')


if __name__ == '__main__':
    main()
