"""Deterministic synthetic weights/inputs shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY.  Weights are drawn from a seeded torch CPU generator in sorted-name
order, so a fixture only has to store {name: shape} + the seed, not the tensors.
"""
import math

import torch


def synth_state_dict(shapes, seed):
    """shapes: {name: tuple}.  1-D '*.weight' -> 1 + 0.1 N(0,1) (norm gains); '*bias' -> 0.1 N(0,1);
    everything else -> N(0,1)/sqrt(fan_in)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        t = torch.randn(shape, generator=g)
        if name.endswith('bias'):
            t = 0.1 * t
        elif len(shape) == 1:
            t = 1.0 + 0.1 * t
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = t / math.sqrt(fan_in)
        sd[name] = t
    return sd


def seeded_input(batch=1, seed=1234):
    """Config-2 input of SURVEY.md 8(d): rand(B,3,512,512, seed 1234)*2-1."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, 3, 512, 512, generator=g) * 2 - 1


def seeded_randn(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def synth_detector_state_dict(shapes, seed):
    """Well-conditioned synthetic weights for the RetinaFace detector, from {name: shape} in sorted-name order (so the reference's
    module and this repo's, whose key sets are equal, get the same tensors): He-scaled convolutions, non-trivial BatchNorm statistics
    and affine parameters; the last BatchNorm of every ResNet bottleneck (`bn3`) gets a small gain so that 16 residual blocks with
    FIXED statistics do not blow the activations up; the stem is scaled for 0..255 inputs and the heads are damped."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if name.endswith('num_batches_tracked'):
            sd[name] = torch.zeros(shape, dtype=torch.int64)
        elif name.endswith('running_var'):
            sd[name] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif name.endswith('running_mean'):
            sd[name] = torch.randn(shape, generator=g) * 0.1
        elif name.endswith('bias'):
            sd[name] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 1:
            lo, hi = (0.2, 0.4) if '.bn3.' in name else (0.75, 1.25)
            sd[name] = torch.rand(shape, generator=g) * (hi - lo) + lo
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 1.0
            if name in ('body.conv1.weight', 'body.stage1.0.0.weight'):
                gain = 1.0 / 64.0          # the stem sees mean-subtracted 0..255 pixels
            elif name.endswith('conv1x1.weight'):
                gain = 0.1                 # heads: unsaturated class probabilities, regression offsets of O(0.1)
            sd[name] = torch.randn(shape, generator=g) * (gain * math.sqrt(2.0 / fan_in))
    return sd


# ---- range-robustness variants of a state_dict ---------------------------------------------------------------
# The un-normalised streams of CodeFormer.forward are the residual stream (fed straight into Upsample.conv), the quantised feature
# (generator block 0), the CFT branch (`encode_enc` output -> scale.0 / shift.0 -> LeakyReLU -> scale.2 / shift.2) and lq_feat (AdaIN
# target, Transformer input).  These keys set their magnitude:
def _stream_keys(sd):
    keys = []
    for k in sd:
        if not (k.endswith('.weight') or k.endswith('.bias')):
            continue
        base = k.rsplit('.', 1)[0]
        if base.endswith('.conv_out') or base.endswith('.conv2') or base.endswith('.proj_out'):
            keys.append(k)                                                     # what a ResBlock / AttnBlock adds to the residual stream
        elif base.startswith('generator.blocks.') and base.endswith('.conv') and sd[base + '.weight'].dim() == 4 \
                and base.count('.') == 3:                                      # Upsample.conv (generator.blocks.N.conv)
            keys.append(k)
        elif base.startswith('fuse_convs_dict.') and ('.scale.' in base or '.shift.' in base):
            keys.append(k)
        elif base == 'encoder.blocks.24':                                      # encoder head: lq_feat (AdaIN statistics, tokens)
            keys.append(k)
    return keys


def range_variant(sd, kind, seed=4321, calib=None):
    """Deterministic modification of a CodeFormer state_dict that stretches the magnitude of the un-normalised streams.
    kind: 'big' (stream-setting weights and biases x 64: un-normalised conv inputs of 1e3..1e9, far beyond the IEEE-half range;
                 SFT is multiplicative -- dec * scale(e), with e growing like dec -- so the LAST conv of every scale / shift branch
                 is divided by `calib[name]`, the power of two oracle/make_golden_range.py measured on that conv's input with the
                 reference, which keeps the reference itself inside fp32; the goldens store the table),
          'small' (x 1/4096: streams of 1e-4..1e-6, below the half normals and, in the CFT branch, near its subnormal step),
          'heavy' (trained-like: every conv / linear weight gets a heavy-tailed per-element factor, norm gains are log-normal,
                   norm biases N(0, 0.3) -- wide dynamic range inside the normalised streams as well)."""
    out = {k: v.clone() for k, v in sd.items()}
    if kind in ('big', 'small'):
        f = 64.0 if kind == 'big' else 1.0 / 4096.0
        for k in _stream_keys(sd):
            if kind == 'big' and ('.scale.2.' in k or '.shift.2.' in k):
                if k.endswith('.weight'):
                    out[k] = out[k] / float((calib or {}).get(k, 1.0))
                continue
            out[k] = out[k] * f
        return out
    if kind == 'heavy':
        g = torch.Generator().manual_seed(seed)
        for k in sorted(sd):
            v = out[k]
            if not v.dtype.is_floating_point:
                continue
            if v.dim() >= 2 and k.endswith('weight') and 'embedding' not in k:
                t = torch.randn(v.shape, generator=g) / torch.sqrt(torch.randn(v.shape, generator=g) ** 2 * 0.5
                                                                   + torch.randn(v.shape, generator=g) ** 2 * 0.5 + 1e-3)
                out[k] = v * (0.5 + 0.5 * t.abs().clamp(max=40.0))           # Student-t(2)-like factor, median ~1, tail to 20x
            elif v.dim() == 1 and ('norm' in k or k.endswith('.23.weight') or k.endswith('.23.bias') or 'idx_pred_layer.0' in k):
                if k.endswith('weight'):
                    out[k] = v * torch.exp(0.5 * torch.randn(v.shape, generator=g))
                else:
                    out[k] = v + 0.3 * torch.randn(v.shape, generator=g)
        return out
    raise ValueError(kind)


def sweep32_inputs(golden_dir):
    """The 32 faces of the round-5 logit sweep (tests/golden/logit_sweep32.npz), rebuilt from fixtures that are already committed:
    the reference's three real crops (uint8 inputs stored in tests/golden/real_*.npz) under eight EXACT transforms each (flips, a
    quarter turn, a cyclic shift, negation, channel reversal, halving and flip + negation: bit-exact on every device) plus eight
    seeded uniform-noise faces.  Natural-image statistics with 24 different token layouts, without new image fixtures.
    Returns a (32, 3, 512, 512) float32 CPU tensor."""
    import os

    import numpy as np
    faces = []
    for name in ('real_0143.npz', 'real_0342.npz', 'real_Solvay_conference_1927_0018.npz'):
        img = np.load(os.path.join(golden_dir, name))['img']                      # uint8 HWC BGR, as cv2.imread returns it
        t = torch.from_numpy(np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1))).float() / 255.   # img2tensor(img / 255., bgr2rgb=True)
        t = (t - 0.5) / 0.5
        faces += [t.flip(2), t.flip(1), t.rot90(1, (1, 2)), t.roll((37, 101), (1, 2)), -t, t.flip(0), t * 0.5, -t.flip(2)]
    faces = [f.contiguous() for f in faces] + list(seeded_input(8, seed=777))
    return torch.stack(faces)
