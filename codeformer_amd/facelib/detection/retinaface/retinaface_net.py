"""RetinaFace building blocks: backbones (MobileNetV1-0.25, ResNet-50 trunk), FPN, SSH context modules and the three 1x1 heads.

Mirror of the reference's facelib/detection/retinaface/retinaface_net.py:7-196 plus the slice of torchvision's ResNet-50 the reference
takes through `IntermediateLayerGetter` (retinaface.py:90-97: everything up to `layer4`; `avgpool` / `fc` are dropped by the getter,
so `detection_Resnet50_Final.pth` holds `body.conv1 ... body.layer4.*` only).  The `state_dict` keys and shapes are the reference's,
so both released checkpoints load with strict=True; the module tree is table-driven instead of spelled out.

Every conv + BatchNorm (+ LeakyReLU) triple is a `ConvUnit`: an `nn.Sequential` whose children carry the reference's indices
(0 conv, 1 bn, 2 act).  On the host it runs as stock torch ops; `folded()` hands the HIP path the BatchNorm-folded weight.
"""
import torch
from torch import nn
from torch.nn import functional as F


class ConvUnit(nn.Sequential):
    """Conv2d(k, stride, pad k//2, bias False) -> BatchNorm2d [-> LeakyReLU(slope)]; slope None = no activation (slope 0 = ReLU)."""

    def __init__(self, cin, cout, k=3, stride=1, slope=0.0, groups=1):
        layers = [nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False), nn.BatchNorm2d(cout)]
        if slope is not None:
            layers.append(nn.LeakyReLU(negative_slope=slope, inplace=True))
        super().__init__(*layers)
        self.slope = slope

    def folded(self):
        """(weight, bias) of the equivalent bias-carrying convolution in eval mode."""
        conv, bn = self[0], self[1]
        g = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        return (conv.weight.detach().float() * g.view(-1, 1, 1, 1)).contiguous(), \
            (bn.bias.detach().float() - bn.running_mean.float() * g).contiguous()


def conv_bn(inp, oup, stride=1, leaky=0):                 # retinaface_net.py:6-9
    return ConvUnit(inp, oup, 3, stride, leaky)


def conv_bn_no_relu(inp, oup, stride):                    # :12-16
    return ConvUnit(inp, oup, 3, stride, None)


def conv_bn1X1(inp, oup, stride, leaky=0):                # :19-22
    return ConvUnit(inp, oup, 1, stride, leaky)


def conv_dw(inp, oup, stride, leaky=0.1):                 # :25-33 depthwise 3x3 + pointwise 1x1, six children 0..5
    dw, pw = ConvUnit(inp, inp, 3, stride, leaky, groups=inp), ConvUnit(inp, oup, 1, 1, leaky)
    return nn.Sequential(*dw, *pw)


def _ctx_slope(out_channel):
    return 0.1 if out_channel <= 64 else 0.0              # :41-44, :73-76


class SSH(nn.Module):
    """Single-stage-headless context module (:36-63): 3x3, 5x5 (two 3x3) and 7x7 (three 3x3) branches, concatenated, ReLU."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        assert out_channel % 4 == 0
        s, q = _ctx_slope(out_channel), out_channel // 4
        self.conv3X3 = conv_bn_no_relu(in_channel, out_channel // 2, stride=1)
        self.conv5X5_1 = conv_bn(in_channel, q, stride=1, leaky=s)
        self.conv5X5_2 = conv_bn_no_relu(q, q, stride=1)
        self.conv7X7_2 = conv_bn(q, q, stride=1, leaky=s)
        self.conv7x7_3 = conv_bn_no_relu(q, q, stride=1)

    def forward(self, x):
        mid = self.conv5X5_1(x)
        return F.relu(torch.cat([self.conv3X3(x), self.conv5X5_2(mid), self.conv7x7_3(self.conv7X7_2(mid))], dim=1))


class FPN(nn.Module):
    """Top-down pyramid over three backbone taps (:66-100): lateral 1x1, nearest upsample to the finer level's size, add, 3x3 merge."""

    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        s = _ctx_slope(out_channels)
        for i, c in enumerate(in_channels_list[:3]):
            setattr(self, f'output{i + 1}', conv_bn1X1(c, out_channels, stride=1, leaky=s))
        self.merge1 = conv_bn(out_channels, out_channels, leaky=s)
        self.merge2 = conv_bn(out_channels, out_channels, leaky=s)

    def forward(self, feats):
        feats = list(feats.values()) if isinstance(feats, dict) else list(feats)
        o1, o2, o3 = self.output1(feats[0]), self.output2(feats[1]), self.output3(feats[2])
        o2 = self.merge2(o2 + F.interpolate(o3, size=[o2.size(2), o2.size(3)], mode='nearest'))
        o1 = self.merge1(o1 + F.interpolate(o2, size=[o1.size(2), o1.size(3)], mode='nearest'))
        return [o1, o2, o3]


class MobileNetV1(nn.Module):
    """Width-0.25 MobileNetV1 (:103-140); (cout, stride) per depthwise-separable block, stages as the reference splits them."""

    STAGES = (((16, 1), (32, 2), (32, 1), (64, 2), (64, 1)),
              ((128, 2), (128, 1), (128, 1), (128, 1), (128, 1), (128, 1)),
              ((256, 2), (256, 1)))

    def __init__(self):
        super().__init__()
        c = 8
        for si, spec in enumerate(self.STAGES):
            blocks = [conv_bn(3, 8, 2, leaky=0.1)] if si == 0 else []
            for cout, stride in spec:
                blocks.append(conv_dw(c, cout, stride))
                c = cout
            setattr(self, f'stage{si + 1}', nn.Sequential(*blocks))
        self.avg = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(256, 1000)

    def forward(self, x):
        x = self.stage3(self.stage2(self.stage1(x)))
        return self.fc(self.avg(x).view(-1, 256))


class Bottleneck(nn.Module):
    """torchvision's ResNet bottleneck (v1.5: the stride sits on the 3x3): 1x1 -> 3x3(stride) -> 1x1(x4), identity or 1x1(stride)
    projection, ReLU after the sum.  Children named conv1/bn1/conv2/bn2/conv3/bn3/downsample as in the checkpoint."""

    expansion = 4

    def __init__(self, cin, width, stride=1):
        super().__init__()
        cout = width * self.expansion
        self.conv1, self.bn1 = nn.Conv2d(cin, width, 1, bias=False), nn.BatchNorm2d(width)
        self.conv2, self.bn2 = nn.Conv2d(width, width, 3, stride, 1, bias=False), nn.BatchNorm2d(width)
        self.conv3, self.bn3 = nn.Conv2d(width, cout, 1, bias=False), nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class ResNet50Trunk(nn.Module):
    """conv1 7x7/2 -> bn1 -> relu -> maxpool 3x3/2 -> layer1..4 ([3, 4, 6, 3] bottlenecks, widths 64..512).  forward returns the
    outputs of layer2 / layer3 / layer4 (strides 8 / 16 / 32; 512 / 1024 / 2048 channels) -- the three taps of cfg_re50's
    `return_layers` (retinaface.py:51-55)."""

    DEPTHS, WIDTHS = (3, 4, 6, 3), (64, 128, 256, 512)

    def __init__(self):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (n, w) in enumerate(zip(self.DEPTHS, self.WIDTHS)):
            blocks = []
            for j in range(n):
                blocks.append(Bottleneck(cin, w, stride=2 if (j == 0 and i > 0) else 1))
                cin = w * Bottleneck.expansion
            setattr(self, f'layer{i + 1}', nn.Sequential(*blocks))
        for m in self.modules():                      # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        x = self.layer1(self.maxpool(self.relu(self.bn1(self.conv1(x)))))
        c3 = self.layer2(x)
        c4 = self.layer3(c3)
        return [c3, c4, self.layer4(c4)]


class MobileNetTrunk(nn.Module):
    """stage1 / stage2 / stage3 of MobileNetV1 as the getter keeps them (`avg` and `fc` come after the last tap and are dropped)."""

    def __init__(self):
        super().__init__()
        full = MobileNetV1()
        self.stage1, self.stage2, self.stage3 = full.stage1, full.stage2, full.stage3

    def forward(self, x):
        c3 = self.stage1(x)
        c4 = self.stage2(c3)
        return [c3, c4, self.stage3(c4)]


class _Head(nn.Module):
    """1x1 conv with `per_anchor` outputs per anchor, returned as (B, H*W*anchors, per_anchor) (:143-183)."""

    per_anchor = 1

    def __init__(self, inchannels=512, num_anchors=3):
        super().__init__()
        self.num_anchors = num_anchors
        self.conv1x1 = nn.Conv2d(inchannels, num_anchors * self.per_anchor, kernel_size=(1, 1), stride=1, padding=0)

    def forward(self, x):
        out = self.conv1x1(x).permute(0, 2, 3, 1).contiguous()
        return out.view(out.shape[0], -1, self.per_anchor)


class ClassHead(_Head):
    per_anchor = 2


class BboxHead(_Head):
    per_anchor = 4


class LandmarkHead(_Head):
    per_anchor = 10


def _heads(cls, fpn_num, inchannels, anchor_num):
    return nn.ModuleList(cls(inchannels, anchor_num) for _ in range(fpn_num))


def make_class_head(fpn_num=3, inchannels=64, anchor_num=2):       # :186-190
    return _heads(ClassHead, fpn_num, inchannels, anchor_num)


def make_bbox_head(fpn_num=3, inchannels=64, anchor_num=2):        # :193-197
    return _heads(BboxHead, fpn_num, inchannels, anchor_num)


def make_landmark_head(fpn_num=3, inchannels=64, anchor_num=2):    # :200-204
    return _heads(LandmarkHead, fpn_num, inchannels, anchor_num)
