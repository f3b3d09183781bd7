"""Does running two half-batches on two streams beat one batch of 16?  (HBM-bound and MFMA-bound kernels of the two halves could overlap.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import codeformer_amd.archs  # noqa
from codeformer_amd.utils.registry import ARCH_REGISTRY
from oracle.synth import seeded_input
torch.manual_seed(0)
net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=['32', '64', '128', '256']).eval().cuda()
x = seeded_input(16).cuda()
xa, xb = x[:8].contiguous(), x[8:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    net(x, w=0.5, adain=True)
def two():
    with torch.cuda.stream(s1):
        net(xa, w=0.5, adain=True)
    with torch.cuda.stream(s2):
        net(xb, w=0.5, adain=True)
for name, fn in (('one batch of 16', one), ('two streams x 8', two), ('one batch of 16', one), ('two streams x 8', two)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f'{name}: {dt * 1e3:.2f} ms = {16 / dt:.1f} faces/s', flush=True)
