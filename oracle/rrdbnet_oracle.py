"""CPU restatement of the Real-ESRGAN generator, functional over a state_dict -- TEST INFRASTRUCTURE ONLY.

Follows basicsr/archs/rrdbnet_arch.py (ResidualDenseBlock.forward :32-39, RRDB.forward :57-62, RRDBNet.forward :103-119)
and pixel_unshuffle (basicsr/archs/arch_util.py:190-206).  Pinned against outputs of the reference's own file imported
in the build container: tests/golden/rrdbnet_*.npz, written by oracle/make_golden_rrdbnet.py
(tests/test_oracle_golden.py::test_rrdbnet_oracle_matches_reference_goldens).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product path never does.
"""
import torch
import torch.nn.functional as F


def pixel_unshuffle(x, s):
    b, c, hs, ws = x.shape
    h, w = hs // s, ws // s
    out = x.new_empty(b, c * s * s, h, w)
    for ci in range(c):
        for dy in range(s):
            for dx in range(s):
                out[:, (ci * s + dy) * s + dx] = x[:, ci, dy::s, dx::s]
    return out


def _conv(sd, name, x):
    return F.conv2d(x, sd[name + '.weight'], sd[name + '.bias'], stride=1, padding=1)


def _lrelu(x):
    return torch.where(x > 0, x, x * 0.2)


def dense_block(sd, prefix, x):
    x1 = _lrelu(_conv(sd, prefix + '.conv1', x))
    x2 = _lrelu(_conv(sd, prefix + '.conv2', torch.cat((x, x1), 1)))
    x3 = _lrelu(_conv(sd, prefix + '.conv3', torch.cat((x, x1, x2), 1)))
    x4 = _lrelu(_conv(sd, prefix + '.conv4', torch.cat((x, x1, x2, x3), 1)))
    x5 = _conv(sd, prefix + '.conv5', torch.cat((x, x1, x2, x3, x4), 1))
    return x5 * 0.2 + x


def rrdb(sd, prefix, x):
    out = x
    for j in (1, 2, 3):
        out = dense_block(sd, f'{prefix}.rdb{j}', out)
    return out * 0.2 + x


def rrdbnet_forward(sd, x, scale):
    """sd: RRDBNet state_dict (fp32 CPU tensors); x: (B,3,H,W); scale in {1,2,4} as in the constructor."""
    num_block = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('body.'))
    with torch.no_grad():
        feat = x
        if scale == 2:
            feat = pixel_unshuffle(x, 2)
        elif scale == 1:
            feat = pixel_unshuffle(x, 4)
        feat = _conv(sd, 'conv_first', feat)
        t = feat
        for i in range(num_block):
            t = rrdb(sd, f'body.{i}', t)
        feat = feat + _conv(sd, 'conv_body', t)
        for name in ('conv_up1', 'conv_up2'):
            up = feat.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)   # nearest x2: src = floor(dst / 2)
            feat = _lrelu(_conv(sd, name, up))
        return _conv(sd, 'conv_last', _lrelu(_conv(sd, 'conv_hr', feat)))
