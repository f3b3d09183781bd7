"""Real-ESRGAN generator (RRDBNet) on the HIP convolution kernel -- the `--bg_upsampler realesrgan` / `--face_upsample`
network of the reference (basicsr/archs/rrdbnet_arch.py:9-119, built at inference_codeformer.py:19-45).

Same constructor, attribute names and `state_dict` keys as the reference, so `RealESRGAN_x2plus.pth` loads unchanged.
SURVEY.md 8(f)4: a "next" row that reuses the hot path's 3x3 kernel; nothing here is used by CodeFormer.forward.

How the dense block maps onto cf_conv2d (no torch.cat, no stand-alone activation or residual pass):
  * x1..x4 of a ResidualDenseBlock live in ONE 4*grow-channel NHWC buffer; conv_k reads cat(x, x1..x_{k-1}) as
    (in0 = x, in1 = growth[..., :32(k-1)]) and writes its 32 channels in place at growth[..., 32(k-1):32k] with the
    LeakyReLU in the epilogue (channel-strided slices, cf_conv_desc.ld_in1 / ld_out);
  * conv5 applies `x5 * 0.2 + x` in its epilogue (CF_EPI_AXPY); the third block of an RRDB also applies the outer
    `out * 0.2 + x` (CF_EPI_AXPY2), both with separately rounded multiply and add like the two ATen ops;
  * nearest-x2 + conv_up{1,2} + LeakyReLU is the folded sub-pixel kernel with the activation in the epilogue;
  * `feat + conv_body(...)` is a residual epilogue; conv_last writes NCHW directly.
Image sizes are arbitrary (edge tiles are masked); there is no tiling requirement on a 288 GB device.
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops
from ..utils.registry import ARCH_REGISTRY
from .hip_module import HipModule

_SLOPE = 0.2      # LeakyReLU slope (rrdbnet_arch.py:27,105)
_RES_SCALE = 0.2  # residual scaling (rrdbnet_arch.py:39,62)


def _unshuffle_host(x, s):
    """(B,C,H*s,W*s) -> (B,C*s*s,H,W), channel (c*s + dy)*s + dx (arch_util.py:190-206)."""
    b, c, hs, ws = x.shape
    if hs % s or ws % s:
        raise ValueError(f'pixel_unshuffle: {hs}x{ws} not divisible by {s}')
    return x.reshape(b, c, hs // s, s, ws // s, s).permute(0, 1, 3, 5, 2, 4).reshape(b, c * s * s, hs // s, ws // s)


class ResidualDenseBlock(HipModule):
    """Five 3x3 convs with dense connections; conv1..4 grow by `num_grow_ch`, conv5 returns to `num_feat`
    (rrdbnet_arch.py:9-39).  Init: stock Conv2d init, then kaiming-normal weights scaled by 0.1 and zero biases
    (arch_util.py:18-36 with scale=0.1) -- same RNG consumption order as the reference."""

    def __init__(self, num_feat=64, num_grow_ch=32):
        super().__init__()
        self.num_feat, self.num_grow_ch = num_feat, num_grow_ch
        for k in range(1, 6):
            setattr(self, f'conv{k}', nn.Conv2d(num_feat + (k - 1) * num_grow_ch, num_grow_ch if k < 5 else num_feat, 3, 1, 1))
        self.lrelu = nn.LeakyReLU(negative_slope=_SLOPE, inplace=True)
        for k in range(1, 6):
            conv = getattr(self, f'conv{k}')
            nn.init.kaiming_normal_(conv.weight)
            conv.weight.data *= 0.1
            conv.bias.data.fill_(0)

    def run_hip(self, x, growth, out, outer=None, f16=False):
        """x: (B,H,W,num_feat) dense; growth: (B,H,W,4*grow) scratch; out: destination (B,H,W,num_feat).
        outer: the enclosing RRDB's input when this is its last block (second residual fused).
        f16: IEEE-half MFMA operands (fp32 accumulate, fp32 tensors)."""
        g = self.num_grow_ch
        for k in range(1, 5):
            ops.conv2d(x, self._pw_conv(f'conv{k}', f16=f16), x2=growth[..., :g * (k - 1)] if k > 1 else None,
                       epilogue=ops.EPI_LEAKY, out=growth[..., g * (k - 1):g * k])
        pw5 = self._pw_conv('conv5', f16=f16)
        if outer is None:
            return ops.conv2d(x, pw5, x2=growth, epilogue=ops.EPI_AXPY, res=x, sft_w=_RES_SCALE, out=out)
        return ops.conv2d(x, pw5, x2=growth, epilogue=ops.EPI_AXPY2, res=x, sft_scale=outer, sft_w=_RES_SCALE, out=out)

    def forward_host(self, x):
        feats = [x]
        for k in range(1, 5):
            feats.append(F.leaky_relu(getattr(self, f'conv{k}')(torch.cat(feats, 1)), _SLOPE))
        return self.conv5(torch.cat(feats, 1)) * _RES_SCALE + x

    def forward(self, x):
        if x.is_cuda:
            with torch.no_grad():
                xh = ops.to_nhwc(x.float())
                growth = xh.new_empty(xh.shape[:3] + (4 * self.num_grow_ch,))
                return ops.to_nchw(self.run_hip(xh, growth, torch.empty_like(xh)))
        return self.forward_host(x)


class RRDB(HipModule):
    """Three dense blocks + scaled outer residual (rrdbnet_arch.py:42-62)."""

    def __init__(self, num_feat, num_grow_ch=32):
        super().__init__()
        self.rdb1 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb2 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb3 = ResidualDenseBlock(num_feat, num_grow_ch)

    def run_hip(self, x, growth, bufs, f16=False):
        """bufs: three (B,H,W,num_feat) buffers, none of them x; returns bufs[2]."""
        a = self.rdb1.run_hip(x, growth, bufs[0], f16=f16)
        b = self.rdb2.run_hip(a, growth, bufs[1], f16=f16)
        return self.rdb3.run_hip(b, growth, bufs[2], outer=x, f16=f16)

    def forward_host(self, x):
        return self.rdb3(self.rdb2(self.rdb1(x))) * _RES_SCALE + x

    def forward(self, x):
        if x.is_cuda:
            with torch.no_grad():
                xh = ops.to_nhwc(x.float())
                growth = xh.new_empty(xh.shape[:3] + (4 * self.rdb1.num_grow_ch,))
                return ops.to_nchw(self.run_hip(xh, growth, [torch.empty_like(xh) for _ in range(3)]))
        return self.forward_host(x)


@ARCH_REGISTRY.register()
class RRDBNet(HipModule):
    """ESRGAN / Real-ESRGAN generator (rrdbnet_arch.py:65-119).  scale 2 / 1 pixel-unshuffle the input by 2 / 4 first,
    so the network always upsamples its feature map x4; forward(x): (B,num_in_ch,H,W) -> (B,num_out_ch,H*scale,W*scale)."""

    def __init__(self, num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
        super().__init__()
        self.scale = scale
        self.num_feat, self.num_grow_ch, self.num_out_ch = num_feat, num_grow_ch, num_out_ch
        if scale == 2:
            num_in_ch = num_in_ch * 4
        elif scale == 1:
            num_in_ch = num_in_ch * 16
        self.conv_first = nn.Conv2d(num_in_ch, num_feat, 3, 1, 1)
        self.body = nn.Sequential(*[RRDB(num_feat=num_feat, num_grow_ch=num_grow_ch) for _ in range(num_block)])
        self.conv_body = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_hr = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_last = nn.Conv2d(num_feat, num_out_ch, 3, 1, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=_SLOPE, inplace=True)
        # 'fp32' (default): exact fp32 MFMA.  'fp16': IEEE-half MFMA operands with fp32 accumulation and fp32 tensors for every
        # 64/32-wide 3x3 conv (conv_first and conv_last stay fp32) -- what `.half()` asks for (the reference's RealESRGANer
        # default on GPUs, inference_codeformer.py:23-27), at higher accuracy than fp16 storage.
        self.precision = os.environ.get('CODEFORMER_HIP_RRDB_PRECISION', 'fp32')

    def half(self):
        """Module.half() of the reference switches to fp16 storage; here it selects the f16-operand kernels and keeps fp32
        parameters and I/O (inputs of any float dtype are accepted, outputs follow the input dtype)."""
        self.precision = 'fp16'
        return self

    def float(self):
        self.precision = 'fp32'
        return super().float()

    def _unshuffle_factor(self):
        return {2: 2, 1: 4}.get(self.scale, 1)

    def forward_hip(self, x):
        if self.num_feat % 16 or self.num_grow_ch % 16:
            raise ValueError('RRDBNet on HIP needs num_feat and num_grow_ch to be multiples of 16')
        if self.num_out_ch > 4:
            raise NotImplementedError('RRDBNet on HIP writes at most 4 output channels (num_out_ch <= 4)')
        if self.precision not in ('fp32', 'fp16'):
            raise ValueError(f"RRDBNet.precision must be 'fp32' or 'fp16', got {self.precision!r}")
        f16 = self.precision == 'fp16'
        if f16 and (self.num_feat % 32 or self.num_grow_ch % 32):
            raise ValueError('fp16 operands need num_feat and num_grow_ch to be multiples of 32')
        t = ops.pixel_unshuffle_nhwc(x.float(), self._unshuffle_factor())
        feat = ops.conv2d(t, self._pw_conv('conv_first'))
        growth = feat.new_empty(feat.shape[:3] + (4 * self.num_grow_ch,))
        pool = [torch.empty_like(feat) for _ in range(4)]
        cur = feat
        for block in self.body:
            cur = block.run_hip(cur, growth, [b for b in pool if b is not cur][:3], f16=f16)
        feat = ops.conv2d(cur, self._pw_conv('conv_body', f16=f16), epilogue=ops.EPI_RESIDUAL, res=feat)
        del growth, pool, cur
        feat = ops.conv2d(feat, self._pw_conv('conv_up1', up2x=True, f16=f16), upsample=True, epilogue=ops.EPI_LEAKY)
        feat = ops.conv2d(feat, self._pw_conv('conv_up2', up2x=True, f16=f16), upsample=True, epilogue=ops.EPI_LEAKY)
        feat = ops.conv2d(feat, self._pw_conv('conv_hr', f16=f16), epilogue=ops.EPI_LEAKY)
        return ops.conv2d(feat, self._pw_conv('conv_last'), out_nchw=True)

    def forward_host(self, x):
        s = self._unshuffle_factor()
        feat = self.conv_first(_unshuffle_host(x, s) if s > 1 else x)
        feat = feat + self.conv_body(self.body(feat))
        feat = F.leaky_relu(self.conv_up1(F.interpolate(feat, scale_factor=2, mode='nearest')), _SLOPE)
        feat = F.leaky_relu(self.conv_up2(F.interpolate(feat, scale_factor=2, mode='nearest')), _SLOPE)
        return self.conv_last(F.leaky_relu(self.conv_hr(feat), _SLOPE))

    def forward(self, x):
        if x.is_cuda:
            with torch.no_grad():
                return self.forward_hip(x).to(x.dtype)
        return self.forward_host(x.float()).to(x.dtype)
