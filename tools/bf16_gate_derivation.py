"""Where the pixel gate of the bf16 leg (bench.py `config3_rank`, tests/test_gpu_real_images.py, tools/gpu_check.py:g_bf16) comes from.

CPU tool (test infrastructure: imports the oracle).  BASELINE configs 3 / 5 name bf16; SURVEY 8(d) asks for "a stated bf16 tolerance".
The tolerance is derived here instead of being picked: the CPU oracle is run at w = 0.7 on the seeded face with the OPERANDS of every
3x3 convolution of generator + fusion blocks that the product's 'bf16' mode puts on bf16 MFMA (cin % 32 == 0, cout % 4 == 0; activations
rounded after GroupNorm / swish / LeakyReLU, weights rounded once; fp32 accumulation, fp32 tensors -- exactly what the kernels do)
rounded to bf16 (round-to-nearest-even), and compared with the reference's committed fp32 output
(tests/golden/restoration_seed0_face0_w0.7.npz, every 4th pixel).  That difference is what bf16 operands cost ANY implementation of this
network with these weights; the gate is 1.5x its maximum and 1.25x its mean (an implementation may differ from this emulation in the
summation order and in rounding folded upsample taps after folding instead of before).

Round 6: the 'bf16' mode also STORES the generator / fusion activations of more than 1024 pixels as bf16 (BASELINE: "bf16 storage + fp32
accumulate"); the emulation gets the same storage points and the gate is re-derived from it.

usage: python tools/bf16_gate_derivation.py        (a few minutes on 8 cores; writes profiles/r06_bf16_gate_derivation.txt)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import codeformer_oracle as O  # noqa: E402
from oracle.synth import seeded_input  # noqa: E402


def build_sd():
    import codeformer_amd.archs  # noqa: F401
    from codeformer_amd.utils.registry import ARCH_REGISTRY
    torch.manual_seed(0)
    net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=['32', '64', '128', '256']).eval()
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def main():
    sd = build_sd()
    x = seeded_input(1)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'restoration_seed0_face0_w0.7.npz'))
    ref = torch.from_numpy(g['out_sub'])
    plain = O.conv
    state = {'on': False, 'n': 0, 'store': False, 'stored': 0}

    def conv_bf16(xx, sd_, p, stride=1, padding=1):
        wt = sd_[p + '.weight']
        if state['on'] and stride == 1 and tuple(wt.shape[2:]) == (3, 3) and wt.shape[1] % 32 == 0 and wt.shape[0] % 4 == 0:
            state['n'] += 1
            return F.conv2d(xx.bfloat16().float(), wt.bfloat16().float(), sd_.get(p + '.bias'), stride=1, padding=padding)
        return plain(xx, sd_, p, stride=stride, padding=padding)

    def q(t):
        """bf16 STORAGE of a generator / fusion activation of more than 1024 pixels per image (round 6: cf_conv_desc.io_bf16): rounded to
        nearest even once, where the producing launch writes it; smaller tensors (16x16, 32x32) stay fp32."""
        if state['on'] and state['store'] and t.shape[-1] * t.shape[-2] > 1024:
            state['stored'] += 1
            return t.bfloat16().float()
        return t

    def res_block_e(x_in, sd_, p):      # vqgan_arch.py:153-164 with the product's storage points: conv1's output, the 1x1 skip, the block output
        h = O.swish(O.group_norm(x_in, sd_, p + '.norm1'))
        h = q(O.conv(h, sd_, p + '.conv1'))
        h = O.swish(O.group_norm(h, sd_, p + '.norm2'))
        h = O.conv(h, sd_, p + '.conv2')                      # (the fp32 accumulator: the residual is added before the one rounding)
        if (p + '.conv_out.weight') in sd_:
            x_in = q(O.conv(x_in, sd_, p + '.conv_out', padding=0))
        return q(h + x_in)

    def fuse_e(enc, dec, w, sd_, p):    # codeformer_arch.py:151-157: bf16 copy of the tap, e, the two LeakyReLU inputs, scale; the SFT combine rounds once
        e = res_block_e(torch.cat([q(enc), dec], dim=1), sd_, p + '.encode_enc')
        scale = q(O.conv(F.leaky_relu(q(O.conv(e, sd_, p + '.scale.0')), 0.2), sd_, p + '.scale.2'))
        shift = O.conv(F.leaky_relu(q(O.conv(e, sd_, p + '.shift.0')), 0.2), sd_, p + '.shift.2')
        return q(dec + w * (dec * scale + shift))

    def upsample_e(xx, sd_, p):         # vqgan_arch.py:134-138: the bf16 copy of the 32x32 input is what the operand rounding gives anyway
        return q(O.conv(F.interpolate(xx, scale_factor=2.0, mode='nearest'), sd_, p + '.conv'))

    gen = O.generator_forward
    orig = (O.res_block, O.fuse_sft, O.upsample)

    def generator_bf16(*a, **k):      # generator + fusion blocks only: encoder and Transformer never run on bf16
        state['on'] = True
        try:
            return gen(*a, **k)
        finally:
            state['on'] = False

    lines = []
    out32, logits32, _ = O.codeformer_forward(x, sd, w=0.7, adain_flag=True)
    d32 = (out32[:, :, ::4, ::4] - ref).abs()
    lines.append(f'oracle fp32 vs reference golden (w=0.7, every 4th pixel): max {float(d32.max()):.3e} mean {float(d32.mean()):.3e}')
    res = {}
    for store in (False, True):
        state.update(n=0, stored=0, store=store)
        O.conv, O.generator_forward = conv_bf16, generator_bf16
        O.res_block, O.fuse_sft, O.upsample = res_block_e, fuse_e, upsample_e     # (q is the identity while `store` is off)
        try:
            out16, logits16, _ = O.codeformer_forward(x, sd, w=0.7, adain_flag=True)
        finally:
            O.conv, O.generator_forward = plain, gen
            O.res_block, O.fuse_sft, O.upsample = orig
        d = (out16[:, :, ::4, ::4] - ref).abs()
        mx, mean = float(d.max()), float(d.mean())
        res[store] = (mx, mean)
        what = (f'bf16-rounded operands in {state["n"]} 3x3 convolutions of generator + fusion AND bf16 storage of the {state["stored"]} activations of more than 1024 pixels '
                f'(rounded once where they are written; fp32 accumulate)') if store else \
            f'bf16-rounded operands in {state["n"]} 3x3 convolutions of generator + fusion (fp32 accumulate, fp32 tensors: the mode of rounds 2-5)'
        lines.append(f'oracle with {what} vs the same golden: max {mx:.4f} mean {mean:.5f}  (output std {float(ref.std()):.3f}; logits bitwise those of the '
                     f'fp32 oracle: {bool(torch.equal(logits16, logits32))})')
    mx, mean = res[True]
    lines.append(f'gate of the storage mode = 1.5 x max, 1.25 x mean of its intrinsic cost: max {1.5 * mx:.3f}  mean {1.25 * mean:.4f}')
    txt = '\n'.join(lines)
    print(txt)
    with open(os.path.join(ROOT, 'profiles', 'r06_bf16_gate_derivation.txt'), 'w') as f:
        f.write(txt + '\n')


if __name__ == '__main__':
    main()
