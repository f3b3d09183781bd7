"""Time RRDBNet (RealESRGAN_x2plus shape: scale 2, 23 blocks) on the HIP path.  GPU box only.
    python tools/rrdb_bench.py [H W [batch [fp32|fp16]]]      # input image size, default 1080 1920 1 fp32
Prints ms per image, nominal TFLOP/s (2*MAC of every conv as the reference executes it, 9 taps for the upsample convs)
and the per-kernel-kind split from ops.PROFILE events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops  # noqa: E402
from codeformer_amd.archs.rrdbnet_arch import RRDBNet  # noqa: E402


def nominal_flops(h, w, scale=2, nf=64, gc=32, nb=23, cin=3, cout=3):
    s = {2: 2, 1: 4}.get(scale, 1)
    lr = (h // s) * (w // s)
    rdb = 9 * 2 * sum((nf + k * gc) * (gc if k < 4 else nf) for k in range(5))
    per_lr = 9 * 2 * (cin * s * s) * nf + nb * 3 * rdb + 9 * 2 * nf * nf * (1 + 4 + 16 + 16) + 16 * 9 * 2 * nf * cout
    return float(lr) * per_lr


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
    b = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    prec = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
    torch.manual_seed(0)
    net = RRDBNet(3, 3, scale=2, num_feat=64, num_block=23, num_grow_ch=32).eval().cuda()
    net.precision = prec
    x = torch.rand(b, 3, h, w, device='cuda')
    for _ in range(2):
        y = net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        y = net(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = nominal_flops(h, w) * b
    print(f'RRDBNet x2 23 blocks ({prec}), input {b}x3x{h}x{w} -> {tuple(y.shape)}: {ms:.2f} ms/call, {ms / b:.2f} ms/image, '
          f'{fl / ms / 1e9:.1f} nominal TFLOP/s, peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB')
    ops.PROFILE = []
    net(x)
    torch.cuda.synchronize()
    agg = {}
    for kind, flops, nbytes, a, c, shape in ops.PROFILE:
        key = (kind, shape[3], shape[4])
        t = agg.setdefault(key, [0, 0.0, 0.0])
        t[0] += 1
        t[1] += a.elapsed_time(c)
        t[2] += flops
    ops.PROFILE = None
    for key, (cnt, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'  {key[0]:10s} cin {key[1]:4d} cout {key[2]:3d}  x{cnt:4d}  {t:8.2f} ms  {f / t / 1e9:7.1f} TFLOP/s (executed)')


if __name__ == '__main__':
    main()
