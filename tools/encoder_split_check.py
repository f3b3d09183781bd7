"""Encoder on split-half operands (CodeFormer.encoder_precision = 'f16x2') against the reference goldens: logits / lq_feat error and
code indices next to the exact-fp32 encoder, on every whole-network golden (real crops, seeded faces, the index sweep), plus the
batch-16 step time of both.  Checker tool -- reads tests/golden only.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import gpu_check as chk  # noqa: E402
from codeformer_amd import ops  # noqa: E402
from oracle.synth import seeded_input  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def main():
    net = chk.build_net().cuda()
    cases = []
    for name in ('real_0143.npz', 'real_0342.npz', 'real_Solvay_conference_1927_0018.npz'):
        g = np.load(os.path.join(GOLD, name))
        cases.append((name, ops.img_u8_to_tensor(torch.from_numpy(g['img']).unsqueeze(0).cuda()), g['logits'], g['idx'], g['gap']))
    g = np.load(os.path.join(GOLD, 'index_sweep_seed2024.npz'))
    x8 = seeded_input(16, seed=2024)[:8].cuda()
    for mode, gmode in (('fp32', 'fp32'), ('f16x2', 'fp32'), ('f16x2', 'f16x2')):
        net.encoder_precision, net.gemm_precision = mode, gmode
        mode = f'encoder {mode}, gemm {gmode}'
        for name, x, rl, ridx, gap in cases:
            logits, lq = net(x, w=0.5, code_only=True)
            l = logits.cpu().numpy()
            err = np.abs(l - rl)
            idx = l.argmax(-1).reshape(-1)
            safe = gap.reshape(-1) >= 1e-5
            bad = int((idx[safe] != ridx.reshape(-1)[safe]).sum())
            # margin: per token, reference gap / (2 * max logit error of that token) -- above 1 the winner cannot flip
            tok_err = err.reshape(-1, err.shape[-1]).max(-1)
            margin = gap.reshape(-1) / np.maximum(2 * tok_err, 1e-30)
            print(f'[{mode}] {name}: logits max {err.max():.2e} mean {err.mean():.2e}  idx mismatches {bad}  min gap {gap.min():.2e}  '
                  f'min margin (safe tokens) {margin[safe].min():.1f}', flush=True)
        logits, _ = net(x8, w=0.5, code_only=True)
        idx = logits.argmax(-1).cpu().numpy()
        safe = g['gap'] >= 1e-5
        print(f'[{mode}] index sweep: mismatches {int((idx[safe] != g["idx"][safe]).sum())} of {int(safe.sum())}', flush=True)
        if 'logits' in g:
            print(f'[{mode}] index sweep logits max err {np.abs(logits.cpu().numpy() - g["logits"]).max():.2e}')
    # fp32-encoder vs split-encoder logits against each other, seeded batch
    x = seeded_input(16).cuda()
    res = {}
    for mode, gmode in (('fp32', 'fp32'), ('f16x2', 'fp32'), ('f16x2', 'f16x2')):
        net.encoder_precision, net.gemm_precision = mode, gmode
        mode = f'encoder {mode}, gemm {gmode}'
        net.precision = 'f16x2'
        for _ in range(3):
            out = net(x, w=0.5, adain=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            out = net(x, w=0.5, adain=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        res[mode] = [o.cpu() for o in out] + [net.last_indices.cpu()]
        print(f'[{mode}] encoder: {dt * 1e3:.2f} ms per 16 faces = {16 / dt:.1f} faces/s', flush=True)
    a, b = res['encoder fp32, gemm fp32'], res['encoder f16x2, gemm f16x2']
    print(f'split (encoder + Transformer GEMMs) vs exact, 16 seeded faces: logits max {float((a[1] - b[1]).abs().max()):.2e}  lq_feat max {float((a[2] - b[2]).abs().max()):.2e}  '
          f'pixels max {float((a[0] - b[0]).abs().max()):.2e}  indices equal {bool(torch.equal(a[3], b[3]))}')


if __name__ == '__main__':
    main()
