"""EXPERIMENT (not a shipped configuration): what Winograd F(4x4,3x3) in the ENCODER would do to logits and code indices, and what it
would buy.  The encoder decides the indices, so the shipped network never runs F(4,3) there; this probe records the numbers behind that
rule on every golden that carries reference logits / indices (seeded face, three real crops, the 8-face index sweep).  GPU box only.
usage: python tools/f43_encoder_probe.py"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codeformer_amd import ops  # noqa: E402
from oracle.synth import seeded_input  # noqa: E402  (test infrastructure: this tool is a checker, not a product path)

GOLD = os.path.join(ROOT, 'tests', 'golden')
spec = importlib.util.spec_from_file_location('gpu_check', os.path.join(ROOT, 'tools', 'gpu_check.py'))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)
net = chk.build_net().cuda()


def faces():
    g = np.load(os.path.join(GOLD, 'restoration_seed0_face0.npz'))
    yield 'seed0_face0', seeded_input(1).cuda(), g['logits'], g['idx'], g['gap']
    for name in ('real_0143.npz', 'real_0342.npz', 'real_Solvay_conference_1927_0018.npz'):
        g = np.load(os.path.join(GOLD, name))
        yield name[:-4], ops.img_u8_to_tensor(torch.from_numpy(g['img']).unsqueeze(0).cuda()), g['logits'], g['idx'], g['gap']
    g = np.load(os.path.join(GOLD, 'index_sweep_seed2024.npz'))
    yield 'sweep_seed2024 (8 faces)', seeded_input(16, seed=2024)[:8].cuda(), None, g['idx'], g['gap']


def step_ms(x, n=5):
    for _ in range(2):
        net(x, w=0.5, adain=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        net(x, w=0.5, adain=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


x16 = seeded_input(16).cuda()
for precision in ('f16x2', 'fp32'):
    net.precision = precision
    base = {}
    for enc in (False, True):
        net.winograd_f43_encoder = enc
        print(f'--- precision {precision}, encoder on {"F(4x4,3x3) where covered" if enc else "F(2x2,3x3) (shipped)"}: {step_ms(x16):.2f} ms per 16 faces')
        for name, x, ref_logits, ref_idx, gap in faces():
            logits, _ = net(x, w=0.5, code_only=True)
            lg = logits.float().cpu().numpy()
            idx = lg.argmax(-1)
            safe = gap >= 1e-5
            msg = f'  {name:28s} indices differ: {int((idx != ref_idx).sum())} of {idx.size} (on tokens with reference gap >= 1e-5: {int((idx[safe] != ref_idx[safe]).sum())})'
            if ref_logits is not None:
                msg += f'  max |logits - reference| {np.abs(lg - ref_logits).max():.2e}'
            if not enc:
                base[name] = lg
            else:
                msg += f'  max |logits - shipped encoder| {np.abs(lg - base[name]).max():.2e}'
            top2 = np.sort(lg, -1)[..., -2:]
            msg += f'  smallest own top-2 gap {float((top2[..., 1] - top2[..., 0]).min()):.2e}'
            print(msg)
net.winograd_f43_encoder = False
