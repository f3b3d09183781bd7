"""ParseNet -- the face-parsing network of the paste-back step (reference: facelib/parsing/parsenet.py:8-194, used at
facelib/utils/face_restoration_helper.py:455-466) -- on the HIP convolution kernel.

Same module tree, constructor arguments and `state_dict` keys as the reference (encoder / body / decoder of
ConvLayer / ResidualBlock, BatchNorm inside `norm.norm`), so `parsing_parsenet.pth` loads unchanged.  SURVEY.md 8(f)3:
a "next" row; nothing here is used by CodeFormer.forward.  Supported on the GPU: the configuration the reference
instantiates (norm 'bn', LeakyReLU; also norm / relu 'none'), eval mode.

Mapping onto cf_conv2d (general instantiations):
  * ReflectionPad2d(1) + unpadded conv = the gather's reflect border mode; `scale='down'` = stride 2 with one padded
    row / column on every side; `scale='up'` (nearest x2, reflection pad, conv) = the folded sub-pixel kernel with an
    edge-clamped footprint (reflecting the upsampled image == replicating the source edge);
  * eval-mode BatchNorm is folded into the packed weights (w * gamma * rstd per output channel, bias = beta - mean * ...):
    same function, different rounding order (parity tolerance in tests/test_parsenet.py);
  * LeakyReLU(0.2) and `identity + res` are conv epilogues; `feat + body(feat)` rides in the last body block's epilogue
    (CF_EPI_AXPY2 with alpha = 1: multiplying by 1 is exact);
  * out_mask_conv is padded from 19 to 20 output channels (vector epilogue); `parse_labels` takes the per-pixel argmax on
    the device with the padded class biased to -inf.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ... import ops
from ...archs.hip_module import HipModule

_SLOPE = 0.2


class NormLayer(nn.Module):
    """parsenet.py:8-38: wraps the normalisation so that its parameters live under `.norm`."""

    def __init__(self, channels, normalize_shape=None, norm_type='bn'):
        super().__init__()
        self.norm_type = norm_type.lower()
        if self.norm_type == 'bn':
            self.norm = nn.BatchNorm2d(channels, affine=True)
        elif self.norm_type == 'none':
            self.norm = nn.Identity()
        else:
            raise NotImplementedError(f'norm type {norm_type} (HIP path: bn | none)')

    def forward(self, x):
        return self.norm(x)


class ReluLayer(nn.Module):
    """parsenet.py:41-71."""

    def __init__(self, channels, relu_type='relu'):
        super().__init__()
        self.relu_type = relu_type.lower()
        if self.relu_type == 'leakyrelu':
            self.func = nn.LeakyReLU(_SLOPE, inplace=True)
        elif self.relu_type == 'none':
            self.func = nn.Identity()
        else:
            raise NotImplementedError(f'relu type {relu_type} (HIP path: leakyrelu | none)')

    def forward(self, x):
        return self.func(x)


class ConvLayer(HipModule):
    """[nearest x2] -> ReflectionPad2d(k//2) -> Conv2d(k, stride 1|2, no padding) -> norm -> activation (parsenet.py:74-111)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, scale='none', norm_type='none', relu_type='none', use_pad=True,
                 bias=True):
        super().__init__()
        self.use_pad, self.norm_type, self.scale = use_pad, norm_type, scale
        if norm_type in ('bn',):
            bias = False
        self.reflection_pad = nn.ReflectionPad2d(int(math.ceil((kernel_size - 1.) / 2)))
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, 2 if scale == 'down' else 1, bias=bias)
        self.relu = ReluLayer(out_channels, relu_type)
        self.norm = NormLayer(out_channels, norm_type=norm_type)

    # -- HIP -----------------------------------------------------------------------------------------------------------------
    def _folded(self, cout_to=None):
        """(weight, bias) with the eval-mode BatchNorm folded in; optionally zero-padded to `cout_to` output channels whose
        bias is -inf-like, so they can never win an argmax."""
        w = self.conv2d.weight.detach().float()
        b = self.conv2d.bias.detach().float() if self.conv2d.bias is not None else torch.zeros(w.shape[0], device=w.device)
        if isinstance(self.norm.norm, nn.BatchNorm2d):
            bn = self.norm.norm
            g = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
            w = w * g.view(-1, 1, 1, 1)
            b = (b - bn.running_mean.float()) * g + bn.bias.detach().float()
        if cout_to is not None and cout_to > w.shape[0]:
            extra = cout_to - w.shape[0]
            w = torch.cat([w, w.new_zeros((extra,) + tuple(w.shape[1:]))], 0)
            b = torch.cat([b, b.new_full((extra,), -3.0e38)], 0)
        return w.contiguous(), b.contiguous()

    def packed(self, cout_to=None):
        bn = self.norm.norm if isinstance(self.norm.norm, nn.BatchNorm2d) else None
        deps = [self.conv2d.weight, self.conv2d.bias] + ([bn.weight, bias_of(bn), bn.running_mean, bn.running_var] if bn else [])

        def build():
            w, b = self._folded(cout_to)
            return ops.pack_weight(w, b, up2x=self.scale == 'up')
        return self._packed(('w', cout_to, self.scale), build, *deps)

    def run_hip(self, x, *, epilogue=None, cout_to=None, **kw):
        if self.training:
            raise RuntimeError('ParseNet on HIP folds BatchNorm: call .eval() first')
        if self.conv2d.kernel_size != (3, 3) or not self.use_pad:
            raise NotImplementedError('HIP path: reflection-padded 3x3 convolutions')
        if epilogue is None:
            epilogue = ops.EPI_LEAKY if self.relu.relu_type == 'leakyrelu' else ops.EPI_NONE
        elif self.relu.relu_type != 'none':
            raise ValueError('a fused residual epilogue needs relu_type none')
        pw = self.packed(cout_to)
        if self.scale == 'up':
            return ops.conv2d(x, pw, upsample=True, pad_mode=ops.PAD_EDGE, epilogue=epilogue, **kw)
        if self.scale == 'down':
            return ops.conv2d(x, pw, stride=2, pad_lo=1, pad_mode=ops.PAD_REFLECT, epilogue=epilogue, **kw)
        return ops.conv2d(x, pw, pad_mode=ops.PAD_REFLECT, epilogue=epilogue, **kw)

    # -- host / module API -----------------------------------------------------------------------------------------------------
    def forward_host(self, x):
        if self.scale == 'up':
            x = F.interpolate(x, scale_factor=2, mode='nearest')
        if self.use_pad:
            x = self.reflection_pad(x)
        return self.relu(self.norm(self.conv2d(x)))

    def forward_nhwc(self, x):
        return self.run_hip(x)


def bias_of(bn):
    return bn.bias


class ResidualBlock(HipModule):
    """identity (or a plain ConvLayer shortcut when the shape changes) + conv2(conv1(x)) (parsenet.py:114-139)."""

    def __init__(self, c_in, c_out, relu_type='prelu', norm_type='bn', scale='none'):
        super().__init__()
        self.has_shortcut = not (scale == 'none' and c_in == c_out)
        if self.has_shortcut:
            self.shortcut_func = ConvLayer(c_in, c_out, 3, scale)
        first, second = {'down': ('none', 'down'), 'up': ('up', 'none'), 'none': ('none', 'none')}[scale]
        self.conv1 = ConvLayer(c_in, c_out, 3, first, norm_type=norm_type, relu_type=relu_type)
        self.conv2 = ConvLayer(c_out, c_out, 3, second, norm_type=norm_type, relu_type='none')

    def run_hip(self, x, outer=None):
        """outer: an extra tensor added to the block's output in the same epilogue (feat + body(feat))."""
        identity = self.shortcut_func.run_hip(x) if self.has_shortcut else x
        h = self.conv1.run_hip(x)
        if outer is None:
            return self.conv2.run_hip(h, epilogue=ops.EPI_RESIDUAL, res=identity)
        return self.conv2.run_hip(h, epilogue=ops.EPI_AXPY2, res=identity, sft_scale=outer, sft_w=1.0)

    def forward_host(self, x):
        identity = self.shortcut_func(x) if self.has_shortcut else x
        return identity + self.conv2(self.conv1(x))

    def forward_nhwc(self, x):
        return self.run_hip(x)


class ParseNet(HipModule):
    """forward(x): (B,3,in,in) RGB in [-1,1] -> (out_mask (B,parsing_ch,out,out), out_img (B,3,out,out)) (parsenet.py:142-194)."""

    def __init__(self, in_size=128, out_size=128, min_feat_size=32, base_ch=64, parsing_ch=19, res_depth=10,
                 relu_type='LeakyReLU', norm_type='bn', ch_range=(32, 256)):
        super().__init__()
        self.res_depth, self.parsing_ch = res_depth, parsing_ch
        act = dict(norm_type=norm_type, relu_type=relu_type)
        lo, hi = ch_range

        def clip(c):
            return max(lo, min(c, hi))

        min_feat_size = min(in_size, min_feat_size)
        down_steps = int(math.log2(in_size // min_feat_size))
        up_steps = int(math.log2(out_size // min_feat_size))
        enc = [ConvLayer(3, base_ch, 3, 1)]
        ch = base_ch                                   # (unclipped running width, as in the reference)
        for _ in range(down_steps):
            enc.append(ResidualBlock(clip(ch), clip(ch * 2), scale='down', **act))
            ch *= 2
        body = [ResidualBlock(clip(ch), clip(ch), **act) for _ in range(res_depth)]
        dec = []
        for _ in range(up_steps):
            dec.append(ResidualBlock(clip(ch), clip(ch // 2), scale='up', **act))
            ch //= 2
        self.encoder, self.body, self.decoder = nn.Sequential(*enc), nn.Sequential(*body), nn.Sequential(*dec)
        self.out_img_conv = ConvLayer(clip(ch), 3)
        self.out_mask_conv = ConvLayer(clip(ch), parsing_ch)

    def _trunk_hip(self, x):
        """(B,3,H,W) NCHW -> decoder features (B,H',W',C) NHWC."""
        t = ops.pixel_unshuffle_nhwc(x.float(), 1)            # NCHW -> NHWC, 3 -> 16 zero-padded channels
        feat = self.encoder[0].run_hip(t)
        for blk in list(self.encoder)[1:]:
            feat = blk.run_hip(feat)
        h = feat
        n = len(self.body)
        for i, blk in enumerate(self.body):
            h = blk.run_hip(h, outer=feat if i == n - 1 else None)
        if n == 0:
            raise NotImplementedError('res_depth = 0')
        for blk in self.decoder:
            h = blk.run_hip(h)
        return h

    def _mask_pad(self):
        return (self.parsing_ch + 3) // 4 * 4

    def forward_hip(self, x):
        h = self._trunk_hip(x)
        img = self.out_img_conv.run_hip(h, out_nchw=True)
        mask = self.out_mask_conv.run_hip(h, cout_to=self._mask_pad())
        return ops.to_nchw(mask)[:, :self.parsing_ch], img

    @torch.no_grad()
    def parse_labels(self, x):
        """argmax over the parsing classes per pixel, (B,H,W) int64 -- what face_restoration_helper.py:464-465 computes on
        the host from out_mask; on CUDA tensors the argmax runs on the device without materialising the NCHW mask."""
        if not x.is_cuda:
            return self.forward_host(x)[0].argmax(dim=1)
        h = self._trunk_hip(x)
        mask = self.out_mask_conv.run_hip(h, cout_to=self._mask_pad())
        B, H, W, C = mask.shape
        return ops.argmax_rows(mask.view(B * H * W, C)).view(B, H, W)

    def forward_host(self, x):
        feat = self.encoder(x)
        x = self.decoder(feat + self.body(feat))
        return self.out_mask_conv(x), self.out_img_conv(x)

    def forward(self, x):
        if x.is_cuda:
            with torch.no_grad():
                return self.forward_hip(x)
        return self.forward_host(x)
