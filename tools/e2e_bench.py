"""End-to-end --has_aligned throughput INCLUDING PNG decode / encode and PCIe: N aligned crops on disk -> N restored PNGs through
codeformer_amd.pipeline.AlignedFacePipeline.  (The headline metric of bench.py starts and ends with tensors resident in HBM; this
is the figure next to it.)  usage: python tools/e2e_bench.py [nfaces] [workers] [batch]"""
import glob
import os
import shutil
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codeformer_amd.pipeline import AlignedFacePipeline  # noqa: E402
from codeformer_amd.utils.img_util import imwrite  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
workers = int(sys.argv[2]) if len(sys.argv) > 2 else None
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
src, dst = '/tmp/e2e_in', '/tmp/e2e_out'
shutil.rmtree(src, ignore_errors=True)
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(src)
crops = [np.load(f)['img'] for f in sorted(glob.glob(os.path.join(ROOT, 'tests/golden/real_*.npz')))]   # the reference's own crops
for i in range(n):
    imwrite(np.roll(crops[i % len(crops)], i // len(crops), axis=1), os.path.join(src, f'{i:05d}.png'))
import codeformer_amd.archs  # noqa: E402,F401
from codeformer_amd.utils.registry import ARCH_REGISTRY  # noqa: E402
torch.manual_seed(0)
net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=['32', '64', '128', '256']).eval().cuda()
paths = sorted(glob.glob(os.path.join(src, '*.png')))
outs = [os.path.join(dst, 'restored_faces', os.path.basename(p)) for p in paths]
pipe = AlignedFacePipeline(net, 'cuda', batch_size=batch, workers=workers)
pipe.restore(paths[:2 * batch], outs[:2 * batch], w=0.5)           # warm-up: weight packing, pool start
st = pipe.restore(paths, outs, w=0.5)
x = torch.rand(batch, 3, 512, 512, device='cuda') * 2 - 1
for _ in range(2):
    net(x, w=0.5, adain=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    net(x, w=0.5, adain=True)
torch.cuda.synchronize()
gpu = 5 * batch / (time.perf_counter() - t0)
print(f'end-to-end: {st["faces"]} faces, batch {batch}, {pipe.workers} I/O workers: {st["faces_per_s"]:.1f} faces/s including PNG decode/encode + PCIe '
      f'(resident-tensor rate on this box {gpu:.1f} faces/s; host waited {st["wait_decode_s"]:.2f} s on decodes, {st["wait_slot_s"]:.2f} s on writes '
      f'of {st["seconds"]:.2f} s)')
