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
