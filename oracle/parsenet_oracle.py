"""CPU restatement of ParseNet (eval mode), functional over a state_dict -- TEST INFRASTRUCTURE ONLY.

Follows facelib/parsing/parsenet.py: ConvLayer.forward :103-111 (nearest x2 -> ReflectionPad2d(1) -> conv -> BatchNorm ->
LeakyReLU 0.2), ResidualBlock.forward :133-139, ParseNet.forward :188-194, for the configuration the reference instantiates
(norm 'bn', relu 'LeakyReLU').  Pinned against outputs of the reference's own file imported in the build container
(tests/golden/parsenet_*.npz, oracle/make_golden_parsenet.py).  Only tests/ may import this module.
"""
import torch
import torch.nn.functional as F


def _reflect_pad1(x):
    """One reflected row / column on every side, written out (no F.pad): index -1 -> 1, n -> n-2."""
    x = torch.cat([x[:, :, 1:2], x, x[:, :, -2:-1]], dim=2)
    return torch.cat([x[:, :, :, 1:2], x, x[:, :, :, -2:-1]], dim=3)


def conv_layer(sd, prefix, x, scale='none'):
    if scale == 'up':
        x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    y = F.conv2d(_reflect_pad1(x), sd[prefix + '.conv2d.weight'], sd.get(prefix + '.conv2d.bias'), stride=2 if scale == 'down' else 1)
    if prefix + '.norm.norm.running_mean' in sd:
        mean, var = sd[prefix + '.norm.norm.running_mean'], sd[prefix + '.norm.norm.running_var']
        y = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
        y = y * sd[prefix + '.norm.norm.weight'].view(1, -1, 1, 1) + sd[prefix + '.norm.norm.bias'].view(1, -1, 1, 1)
        return y, True
    return y, False


def residual_block(sd, prefix, x, scale='none'):
    first, second = {'down': ('none', 'down'), 'up': ('up', 'none'), 'none': ('none', 'none')}[scale]
    identity = conv_layer(sd, prefix + '.shortcut_func', x, scale)[0] if prefix + '.shortcut_func.conv2d.weight' in sd else x
    h, _ = conv_layer(sd, prefix + '.conv1', x, first)
    h = torch.where(h > 0, h, h * 0.2)
    h, _ = conv_layer(sd, prefix + '.conv2', h, second)
    return identity + h


def parsenet_forward(sd, x):
    """sd: ParseNet state_dict (CPU fp32); x: (B,3,H,W).  Returns (out_mask, out_img)."""
    count = lambda stem: 1 + max(int(k.split('.')[1]) for k in sd if k.startswith(stem + '.'))   # noqa: E731
    with torch.no_grad():
        feat, _ = conv_layer(sd, 'encoder.0', x)
        for i in range(1, count('encoder')):
            feat = residual_block(sd, f'encoder.{i}', feat, 'down')
        h = feat
        for i in range(count('body')):
            h = residual_block(sd, f'body.{i}', h)
        h = feat + h
        for i in range(count('decoder')):
            h = residual_block(sd, f'decoder.{i}', h, 'up')
        return conv_layer(sd, 'out_mask_conv', h)[0], conv_layer(sd, 'out_img_conv', h)[0]
