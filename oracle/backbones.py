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
"""

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
    block, depths = _CFG[name]
    return ResNet(block, depths)


def names():
    return sorted(_CFG)
