"""BASELINE config 4 as a synthetic bench: N 1080p frames with k faces each (host-side detection replaced by given alignment
matrices) through codeformer_amd.video.VideoRestorer -- crops cut on the device, faces of many frames batched into 16-face
forwards, paste-back on the device at --upscale.  Reports faces/s and frames/s with the frames starting and ending in HOST memory
(PCIe included) and with the network alone for comparison.  With a fourth argument (cuda | cpu) the detection stage of
FaceRestoreHelper.get_face_landmarks_5(resize=640) -- INTER_AREA reduction on the host, RetinaFace-ResNet50 (seeded weights: its boxes are
discarded, the prepared matrices are used) on that device -- runs per frame inside the timed region.
usage: python tools/video_bench.py [frames] [faces_per_frame] [upscale] [detector device]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import codeformer_amd.archs  # noqa: E402,F401
from codeformer_amd.utils.registry import ARCH_REGISTRY  # noqa: E402
from codeformer_amd.video import VideoRestorer  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 300
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
up = int(sys.argv[3]) if len(sys.argv) > 3 else 2
det_dev = sys.argv[4] if len(sys.argv) > 4 else None
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
frames = [np.roll(base, 7 * i, axis=1) for i in range(nf)]


def affine(cx, cy, size, ang):
    s = 512.0 / size
    c, sn = np.cos(ang) * s, np.sin(ang) * s
    return np.array([[c, -sn, 256 - (c * cx - sn * cy)], [sn, c, 256 - (sn * cx + c * cy)]])


affs = [np.stack([affine(rng.uniform(200, 1700), rng.uniform(200, 880), rng.uniform(120, 320), rng.uniform(-0.4, 0.4)) for _ in range(k)])
        for _ in range(nf)]
torch.manual_seed(0)
net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=['32', '64', '128', '256']).eval().cuda()
vr = VideoRestorer(net, 'cuda', upscale=up, batch_size=16)
vr.restore(frames[:12], affs[:12])            # warm-up (weight packing, allocator)
torch.cuda.synchronize()
det_note = ''
t0 = time.perf_counter()
if det_dev:
    from codeformer_amd import ops
    from codeformer_amd.facelib.detection import RetinaFace
    from codeformer_amd.utils.img_util import resize_area
    det = RetinaFace('resnet50', device=det_dev)

    def detect(f):
        if det_dev == 'cuda':       # the frame goes up once; the INTER_AREA reduction is a kernel, the detector reads its output
            return det.detect_faces(ops.resize_area_u8(torch.from_numpy(f).cuda(), 640, 1137), conf_threshold=2.0)
        return det.detect_faces(resize_area(f, (1137, 640)), conf_threshold=2.0)

    detect(frames[0])      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for f in frames:
            detect(f)
    torch.cuda.synchronize()
    det_note = f'; detection stage (INTER_AREA reduction + RetinaFace-ResNet50 on {det_dev}) {nf / (time.perf_counter() - t0):.1f} frames/s, included'
out = vr.restore(frames, affs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
x = torch.rand(16, 3, 512, 512, device='cuda') * 2 - 1
for _ in range(2):
    net(x, w=0.5, adain=True)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(5):
    net(x, w=0.5, adain=True)
torch.cuda.synchronize()
gpu = 80 / (time.perf_counter() - t1)
print(f'video fan-out: {nf} frames 1920x1080 -> {out[0].shape[1]}x{out[0].shape[0]}, {k} faces/frame, {vr.stats["forward_calls"]} forward calls of 16: '
      f'{nf * k / dt:.1f} faces/s = {nf / dt:.1f} frames/s (frames from / to host memory; network alone on resident tensors {gpu:.1f} faces/s{det_note})')
