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
