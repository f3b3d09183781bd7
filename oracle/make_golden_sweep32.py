"""Golden of the round-5 logit sweep from the REFERENCE's own arch files (build container only; output committed).

TEST INFRASTRUCTURE ONLY.  Run:  python -m oracle.make_golden_sweep32
Needs /root/reference (absent on the GPU box).

The code indices hang on the encoder + Transformer logits, so every change of an encoder kernel is gated on the distance of OUR logits
to the reference's relative to the reference's own top-2 gaps.  The goldens with full reference logits cover nine faces; this one adds
32 more (oracle.synth.sweep32_inputs: the three real crops under eight exact transforms + eight noise faces) at a size that can be
committed: per token the reference's TOP-8 logits (values + code indices), i.e. every code that could plausibly take over the argmax,
and the top-2 gap.  tests/test_gpu_real_images.py::test_encoder_logit_margin reads it.
"""
import os

import numpy as np
import torch

from oracle import ref_loader
from oracle.synth import sweep32_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def main():
    torch.set_num_threads(os.cpu_count())
    reg, vq, cf, _ = ref_loader.load_reference()
    torch.manual_seed(0)
    net = reg.ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                              connect_list=['32', '64', '128', '256']).eval()
    x = sweep32_inputs(GOLD)
    vals, idxs = [], []
    with torch.no_grad():
        for b in range(x.shape[0]):
            logits, _ = net(x[b:b + 1], w=0.5, code_only=True)      # the reference's call pattern: one face per forward
            v, i = torch.topk(logits[0], 9, dim=-1)
            vals.append(v)
            idxs.append(i)
            print(f'face {b}: top-2 gap min {float((v[:, 0] - v[:, 1]).min()):.3e}  distinct codes {int(i[:, 0].unique().numel())}', flush=True)
    v, i = torch.stack(vals), torch.stack(idxs)
    np.savez_compressed(os.path.join(GOLD, 'logit_sweep32.npz'), top_val=v[..., :8].numpy(), top_idx=i[..., :8].numpy().astype(np.int16),
                        ninth_val=v[..., 8].numpy(), gap=(v[..., 0] - v[..., 1]).numpy())
    print('tokens with gap < 1e-5:', int(((v[..., 0] - v[..., 1]) < 1e-5).sum()), 'of', v.shape[0] * v.shape[1])


if __name__ == '__main__':
    main()
