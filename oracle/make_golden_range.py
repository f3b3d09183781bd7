"""Range-robustness goldens: the REFERENCE's CodeFormer.forward with weights that drive the un-normalised
streams far outside / below the IEEE-half range, and with trained-like heavy-tailed weights (build container only).

TEST INFRASTRUCTURE ONLY.  Run:  python -m oracle.make_golden_range        (needs /root/reference; outputs are committed)

Per variant ('big', 'small', 'heavy' of oracle/synth.range_variant, applied to the seed-0 weights) and input (the seeded config-2
face; the reference's inputs/cropped_faces/0143.png for 'heavy'): logits, code indices, top-2 gaps, lq_feat statistics, the
4x-subsampled output and its scale, plus the per-layer max |input| of every 3x3 convolution whose input is NOT normalised
(what the split-half kernels have to carry).  tests/test_gpu_range.py drives the HIP path with the same weights.
"""
import json
import math
import os
import sys

import numpy as np
import torch

from oracle import codeformer_oracle as O
from oracle import ref_loader
from oracle.synth import range_variant, seeded_input

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
LEVELS = ('32', '64', '128', '256')


def _forward(net, x, w, watch=None):
    mags, hooks = {}, []
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and (watch is None or name in watch):
            hooks.append(m.register_forward_hook(lambda mod, i, o, name=name: mags.__setitem__(name, float(i[0].abs().max()))))
    with torch.no_grad():
        out = net(x, w=w, adain=True)
    for h in hooks:
        h.remove()
    return out, mags


def calibrate_big(net, sd0, x, w):
    """Power-of-two divisors for the last conv of every CFT scale / shift branch: max |input| of that conv in the reference,
    level by level (a level's input depends on the levels before it)."""
    calib = {}
    for lv in LEVELS:
        net.load_state_dict(range_variant(sd0, 'big', calib=calib))
        names = [f'fuse_convs_dict.{lv}.{br}.2' for br in ('scale', 'shift')]
        _, mags = _forward(net, x, w, watch=set(names))
        for n in names:
            calib[n + '.weight'] = 2.0 ** math.ceil(math.log2(max(mags[n], 1.0)))
    return calib


def main():
    torch.set_num_threads(os.cpu_count())
    reg, vq, cf, _ = ref_loader.load_reference()
    torch.manual_seed(0)
    net = reg.ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                              connect_list=list(LEVELS)).eval()
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x_seed = seeded_input(1)
    real = np.load(os.path.join(GOLD, 'real_0143.npz'))
    # the reference's boundary (img_util.py:9-35 img2tensor(img / 255., bgr2rgb=True, float32=True), then normalize(0.5, 0.5))
    rgb = torch.from_numpy(np.ascontiguousarray((real['img'] / 255.)[:, :, ::-1].transpose(2, 0, 1))).float()
    x_real = ((rgb - 0.5) / 0.5).unsqueeze(0)
    report = {}
    for kind in ('big', 'small', 'heavy'):
        calib = calibrate_big(net, sd0, x_seed, 0.5) if kind == 'big' else None
        sd = range_variant(sd0, kind, calib=calib)
        net.load_state_dict(sd)
        inputs = [('seed', x_seed)] + ([('real0143', x_real)] if kind == 'heavy' else [])
        for tag, x in inputs:
            (out, logits, lq), mags = _forward(net, x, 0.5)
            assert bool(torch.isfinite(out).all()), (kind, tag)
            unnorm = {k: v for k, v in mags.items() if k.endswith('.conv') and k.startswith('generator') or '.scale.' in k
                      or '.shift.' in k or k == 'generator.blocks.0'}
            top2 = torch.topk(logits, 2, dim=-1).values
            gap = (top2[..., 0] - top2[..., 1])
            o = O.codeformer_forward(x, sd, w=0.5, adain_flag=True, return_idx=True)
            name = f'range_{kind}_{tag}'
            report[name] = {
                'out_absmax': float(out.abs().max()), 'out_absmean': float(out.abs().mean()),
                'logits_absmax': float(logits.abs().max()), 'lq_absmax': float(lq.abs().max()),
                'min_gap': float(gap.min()), 'distinct_codes': int(logits.argmax(-1).unique().numel()),
                'unnormalised_input_absmax': unnorm,
                'oracle_vs_ref_out': float((o[0] - out).abs().max()), 'oracle_vs_ref_logits': float((o[1] - logits).abs().max()),
                'oracle_idx_equal': bool(torch.equal(o[3], logits.argmax(-1))),
            }
            print(name, json.dumps(report[name])[:400], flush=True)
            np.savez_compressed(
                os.path.join(GOLD, name + '.npz'), out_sub=out[:, :, ::4, ::4].contiguous().numpy(), logits=logits.numpy(),
                idx=logits.argmax(-1).numpy(), gap=gap.numpy(), lq_sub=lq[:, ::8].contiguous().numpy(),
                out_absmax=np.float32(out.abs().max()), lq_absmax=np.float32(lq.abs().max()),
                calib_keys=np.array(sorted(calib) if calib else [], dtype='U64'),
                calib_vals=np.array([calib[k] for k in sorted(calib)] if calib else [], dtype=np.float64))
    with open(os.path.join(GOLD, 'range_oracle_vs_reference.json'), 'w') as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print('done')


if __name__ == '__main__':
    sys.exit(main())
