"""ORACLE (test infrastructure, never shipped on the product path).

torch.nn restatement of the torchvision ResNet family that the reference
reaches through ``getattr(torchvision.models, name)(weights=...)``
(``/root/reference/sykepic/train/network.py:48``).  torchvision (pinned
``torchvision==0.12.0`` in ``/root/reference/requirements/cpu.txt:344``; the
code needs >=0.13 for ``weights=``) is a third-party dependency that is not
vendored under ``/root/reference`` and is not installed in this image, so the
published architecture is restated: child order
``conv1,bn1,relu,maxpool,layer1,layer2,layer3,layer4,avgpool,fc`` and
attribute names are what define the ``state_dict`` keys of ``best_state.pth``.
ResNet v1.5 (stride on the 3x3 conv of a Bottleneck).  fp32, NCHW, CPU.

EfficientNet (B0-B4 scaling of the 2019 paper as torchvision >= 0.13 builds it:
``features[0]`` stem 3x3/2, ``features[1..7]`` MBConv stages with
squeeze-excitation, ``features[8]`` 1x1 head conv, ``avgpool``,
``classifier = [Dropout, Linear]``; ``Conv2dNormActivation`` =
``Sequential(Conv2d(bias=False), BatchNorm2d, SiLU)``;
``SqueezeExcitation`` = avgpool -> ``fc1`` (1x1 conv, bias) -> SiLU -> ``fc2``
-> Sigmoid -> scale; stochastic depth is the identity in eval mode).  Checked
against the published parameter counts (efficientnet_b4: 19,341,616 with the
1000-class classifier, efficientnet_b0: 5,288,548).
"""

import math

import torch.nn as nn


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride, pad, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, block, depths, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = _conv(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, depths[0], 1)
        self.layer2 = self._stage(block, 128, depths[1], 2)
        self.layer3 = self._stage(block, 256, depths[2], 2)
        self.layer4 = self._stage(block, 512, depths[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _stage(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(
                _conv(self.inplanes, planes * block.expansion, 1, stride),
                nn.BatchNorm2d(planes * block.expansion),
            )
        blocks = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


_BN = {"eps": 1e-5, "momentum": 0.1}   # norm_layer of the EfficientNet being built (b5-b7: eps 1e-3, momentum 0.01)


def _cna(cin, cout, k, stride=1, groups=1, act=True):
    """torchvision.ops.Conv2dNormActivation(norm=BatchNorm2d, activation=SiLU or None)."""
    mods = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout, **_BN)]
    if act:
        mods.append(nn.SiLU(inplace=True))
    return nn.Sequential(*mods)


class SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.SiLU(inplace=True)
        self.scale_activation = nn.Sigmoid()

    def forward(self, x):
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return s * x


class MBConv(nn.Module):
    def __init__(self, cin, cout, k, stride, expand_ratio):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        exp = _make_divisible(cin * expand_ratio)
        layers = []
        if exp != cin:
            layers.append(_cna(cin, exp, 1))
        layers.append(_cna(exp, exp, k, stride, groups=exp))
        layers.append(SqueezeExcitation(exp, max(1, cin // 4)))
        layers.append(_cna(exp, cout, 1, act=False))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        y = self.block(x)
        return y + x if self.use_res_connect else y   # StochasticDepth: identity in eval


class EfficientNet(nn.Module):
    # expand ratio, kernel, stride, input channels, output channels, layers (B0 baseline)
    _BASE = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3),
             (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))

    def __init__(self, width_mult, depth_mult, dropout, num_classes=1000):
        super().__init__()
        ch = lambda c: _make_divisible(c * width_mult)
        feats = [_cna(3, ch(32), 3, 2)]
        for t, k, s, cin, cout, n in self._BASE:
            cin, cout, n = ch(cin), ch(cout), int(math.ceil(n * depth_mult))
            feats.append(nn.Sequential(*[MBConv(cin if i == 0 else cout, cout, k, s if i == 0 else 1, t)
                                         for i in range(n)]))
        last_in = ch(320)
        feats.append(_cna(last_in, 4 * last_in, 1))
        self.features = nn.Sequential(*feats)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(dropout, inplace=True), nn.Linear(4 * last_in, num_classes))

    def forward(self, x):
        return self.classifier(self.avgpool(self.features(x)).flatten(1))


_EFF = {  # width, depth, dropout (torchvision efficientnet_b0..b7; b5-b7 are built with BatchNorm2d(eps=1e-3, momentum=0.01))
    "efficientnet_b0": (1.0, 1.0, 0.2),
    "efficientnet_b1": (1.0, 1.1, 0.2),
    "efficientnet_b2": (1.1, 1.2, 0.3),
    "efficientnet_b3": (1.2, 1.4, 0.3),
    "efficientnet_b4": (1.4, 1.8, 0.4),
    "efficientnet_b5": (1.6, 2.2, 0.4),
    "efficientnet_b6": (1.8, 2.6, 0.5),
    "efficientnet_b7": (2.0, 3.1, 0.5),
}

_CFG = {
    "resnet18": (BasicBlock, (2, 2, 2, 2)),
    "resnet34": (BasicBlock, (3, 4, 6, 3)),
    "resnet50": (Bottleneck, (3, 4, 6, 3)),
    "resnet101": (Bottleneck, (3, 4, 23, 3)),
    "resnet152": (Bottleneck, (3, 8, 36, 3)),
}


def make(name, weights=None):
    """Stand-in for ``torchvision.models.<name>(weights=...)``; pretrained
    weights cannot be fetched (no network), so ``weights`` is ignored."""
    if name in _EFF:
        big = name in ("efficientnet_b5", "efficientnet_b6", "efficientnet_b7")
        _BN.update(eps=1e-3 if big else 1e-5, momentum=0.01 if big else 0.1)
        try:
            return EfficientNet(*_EFF[name])
        finally:
            _BN.update(eps=1e-5, momentum=0.1)
    block, depths = _CFG[name]
    return ResNet(block, depths)


def names():
    return sorted(_CFG) + sorted(_EFF)
