"""RetinaFace (host-side code, stock torch ops) on the CPU and -- as the reference runs it -- on the ROCm device through MIOpen:
agreement of the two and frames/s at the detector's working size (short side 640, as get_face_landmarks_5(resize=640) feeds it)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facelib.detection.retinaface.retinaface import RetinaFace
from oracle.make_golden_retinaface import build, seeded_frames

torch.set_num_threads(min(64, os.cpu_count()))
for name, seed in (('resnet50', 5), ('mobile0.25', 6)):
    case = dict(network=name, seed=seed)
    cpu = build(RetinaFace, case)
    frames = seeded_frames((4, 640, 1138, 3), 77)                 # 1080p reduced to short side 640
    x = torch.from_numpy(frames.astype(np.float32)).permute(0, 3, 1, 2) - cpu.mean_tensor
    with torch.no_grad():
        t0 = time.time(); ref = cpu(x[:1]); t_cpu = time.time() - t0
        t0 = time.time(); ref = cpu(x[:1]); t_cpu = min(t_cpu, time.time() - t0)
    print(f'{name}: CPU ({torch.get_num_threads()} threads) {t_cpu * 1e3:.0f} ms per 640x1138 frame = {1 / t_cpu:.1f} frames/s', flush=True)
    if not torch.cuda.is_available():
        continue
    try:
        dev = build(RetinaFace, case).to('cuda')
        xd = x.cuda()
        with torch.no_grad():
            out = dev(xd[:1]); torch.cuda.synchronize()
            err = [float((a.cpu() - b).abs().max()) for a, b in zip(out, ref)]
            for bs in (1, 4):
                dev(xd[:bs]); torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(5):
                    dev(xd[:bs])
                torch.cuda.synchronize()
                dt = (time.time() - t0) / 5
                print(f'{name}: ROCm device (torch / MIOpen) batch {bs}: {dt * 1e3:.1f} ms = {bs / dt:.1f} frames/s; max |device - CPU| loc / conf / landmarks = '
                      + ' / '.join(f'{e:.2e}' for e in err), flush=True)
    except Exception as e:                                           # MIOpen may be unusable on a box without its kernel database
        print(f'{name}: ROCm device run failed: {type(e).__name__}: {e}', flush=True)
