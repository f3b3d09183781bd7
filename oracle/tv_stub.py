"""Stand-ins for the slice of torchvision the reference's RetinaFace detector imports (facelib/detection/retinaface/retinaface.py:7,
91-92; retinaface_utils.py:3,41-45).  TEST INFRASTRUCTURE ONLY.

torchvision is a third-party dependency that is absent from /root/reference AND from this image; the reference pins no version
(requirements.txt: `torchvision`).  Restated here from its published definitions:
  * `models.resnet50(pretrained=False)`: ResNet-50 v1.5 -- 7x7/2 stem, 3x3/2 max-pool, stages of [3, 4, 6, 3] bottlenecks
    (1x1 -> 3x3 carrying the stride -> 1x1 with 4x expansion; 1x1 projection shortcut when stride or width changes; ReLU after the
    sum), global average pool, fc 2048 -> 1000;
  * `models._utils.IntermediateLayerGetter(model, return_layers)`: the model's children in registration order up to the last
    requested one, forward returns an OrderedDict {new_name: output};
  * `ops.nms(boxes, scores, iou_threshold)`: greedy suppression by descending score, IoU without +1, strict `>` test.
"""
import types
from collections import OrderedDict

import torch
from torch import nn


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class _ResNet(nn.Module):

    def __init__(self, layers=(3, 4, 6, 3), num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * 4, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50(pretrained=False, **kw):
    assert not pretrained
    return _ResNet()


class IntermediateLayerGetter(nn.ModuleDict):

    def __init__(self, model, return_layers):
        if not set(return_layers).issubset([name for name, _ in model.named_children()]):
            raise ValueError('return_layers are not present in model')
        orig = {str(k): str(v) for k, v in return_layers.items()}
        todo = dict(orig)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            todo.pop(name, None)
            if not todo:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def nms(boxes, scores, iou_threshold):
    boxes, scores = boxes.float(), scores.float()
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    order = scores.argsort(descending=True, stable=True)
    n = boxes.shape[0]
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for a in range(n):
        i = int(order[a])
        if dead[i]:
            continue
        keep.append(i)
        for b in range(a + 1, n):
            j = int(order[b])
            if dead[j]:
                continue
            w = max(0.0, float(min(x2[i], x2[j]) - max(x1[i], x1[j])))
            h = max(0.0, float(min(y2[i], y2[j]) - max(y1[i], y1[j])))
            inter = torch.tensor(w, dtype=torch.float32) * torch.tensor(h, dtype=torch.float32)
            if float(inter / (areas[i] + areas[j] - inter)) > iou_threshold:
                dead[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def modules():
    """{module name: stub module} to install into sys.modules while the reference's detector files are imported."""
    tv = types.ModuleType('torchvision')
    tv.__version__ = '0.0.0-oracle-stub'
    models = types.ModuleType('torchvision.models')
    models.resnet50 = resnet50
    utils = types.ModuleType('torchvision.models._utils')
    utils.IntermediateLayerGetter = IntermediateLayerGetter
    models._utils = utils
    ops = types.ModuleType('torchvision.ops')
    ops.nms = nms
    tv.models, tv.ops = models, ops
    return {'torchvision': tv, 'torchvision.models': models, 'torchvision.models._utils': utils, 'torchvision.ops': ops}
