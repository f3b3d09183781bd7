"""Python surface of the reference's two bundled CUDA extensions, on the HIP C ABI (inference only).

Mirrors basicsr/ops/fused_act/fused_act.py:74-89 (`FusedLeakyReLU`, `fused_leaky_relu`) and
basicsr/ops/upfirdn2d/upfirdn2d.py:147-154 (`upfirdn2d`).  Neither op is used by CodeFormer.forward;
they exist so code written against `basicsr.ops` imports and runs.  GPU tensors always go through
`cf_fused_bias_act_ex` / `cf_upfirdn2d`; the reference itself routes CPU tensors of upfirdn2d to a torch formula
(upfirdn2d.py:148-149) and so does this module.  fused_leaky_relu is differentiable (twice) through the op's own grad modes, exactly as
the reference's autograd Functions are; upfirdn2d is inference-only (a tensor that requires grad raises).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def _no_grad_only(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError('codeformer_amd upfirdn2d is inference-only: call under torch.no_grad()')


class _FusedLeakyReLUBackward(torch.autograd.Function):
    """fused_act.py:25-51: grad_input = op(grad_output, no bias, ref = out, act 3, grad 1); grad_bias = its sum over all but dim 1;
    the double backward is the same op applied to (gradgrad_input, gradgrad_bias)."""

    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = ops.fused_bias_act(grad_output, None, negative_slope, scale, ref=out, act=3, grad=1)
        dim = [0] + list(range(2, grad_input.ndim))
        return grad_input, grad_input.sum(dim).detach()

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        return ops.fused_bias_act(gradgrad_input, gradgrad_bias, ctx.negative_slope, ctx.scale, ref=out, act=3, grad=1), None, None, None


class _FusedLeakyReLU(torch.autograd.Function):
    """fused_act.py:54-71."""

    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = ops.fused_bias_act(input, bias, negative_slope, scale, act=3, grad=0)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = _FusedLeakyReLUBackward.apply(grad_output.contiguous(), out, ctx.negative_slope, ctx.scale)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=math.sqrt(2.0)):
    """leaky_relu(input + bias[channel]) * scale; channel = dim 1 (fused_bias_act_kernel.cu:20-50, act=3), differentiable twice through
    the op's grad modes like the reference's FusedLeakyReLUFunction; float32 / float16 / bfloat16."""
    if input.device.type != 'cuda':
        raise RuntimeError('fused_leaky_relu has no CPU implementation (as in the reference)')
    return _FusedLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """Same constructor and parameter (`bias`, zeros) as fused_act.py:74-86."""

    def __init__(self, channel, negative_slope=0.2, scale=math.sqrt(2.0)):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def _upfirdn2d_host(x, kernel, up, down, pad):
    """Zero-stuff by `up`, pad, correlate with the flipped kernel, keep every `down`-th sample (CPU tensors)."""
    n, c, h, w = x.shape
    z = x.new_zeros(n * c, 1, h * up, w * up)
    z[:, :, ::up, ::up] = x.reshape(n * c, 1, h, w)
    p0, p1 = pad
    z = F.pad(z, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    z = z[:, :, max(-p0, 0):z.shape[2] - max(-p1, 0), max(-p0, 0):z.shape[3] - max(-p1, 0)]
    y = F.conv2d(z, torch.flip(kernel, [0, 1])[None, None].to(z.dtype))
    y = y[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """input (N,C,H,W), kernel (kh,kw): upsample-FIR-downsample with one factor / padding pair for both axes."""
    _no_grad_only(input)
    if input.device.type == 'cpu':
        return _upfirdn2d_host(input, kernel, up, down, pad)
    return ops.upfirdn2d(input, kernel, up=up, down=down, pad=pad)
