"""Image encoder of the fusion models (interface of models/image_encoders.py:28-37 of the reference).

The reference takes torchvision's resnet18 and keeps `children()[:-3]` (conv1, bn1, relu, maxpool,
layer1, layer2, layer3): a 3x224x224 render -> [B, 256, 14*14].  torchvision is not installed
here, so the same network is written out in plain torch with torchvision's module structure, which
makes the state-dict keys identical (`base.0.weight`, `base.1.*`, `base.{4,5,6}.{0,1}.{conv1,bn1,
conv2,bn2}`, `base.{5,6}.0.downsample.{0,1}`) and upstream checkpoints loadable.  Convolutions run
on MIOpen; there is no hand-written kernel here.  ConvNext / ResNet50 variants
are not used by any registered model.
"""
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


def _make_layer(inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes))
    layers = [BasicBlock(inplanes, planes, stride, downsample)]
    layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def resnet18_children(upto_layer=4, num_classes=1000):
    """The child modules of torchvision.models.resnet18, in order, with torchvision's init."""
    mods = [nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            _make_layer(64, 64, 2, 1), _make_layer(64, 128, 2, 2), _make_layer(128, 256, 2, 2),
            _make_layer(256, 512, 2, 2), nn.AdaptiveAvgPool2d((1, 1)), nn.Linear(512, num_classes)]
    for m in mods:
        for sub in m.modules():
            if isinstance(sub, nn.Conv2d):
                nn.init.kaiming_normal_(sub.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(sub, nn.BatchNorm2d):
                nn.init.constant_(sub.weight, 1)
                nn.init.constant_(sub.bias, 0)
    return mods[:4 + upto_layer]


class ImageEncoder(nn.Module):
    def forward(self, x):
        raise NotImplementedError("Subclass must implement forward method.")


class ResNet18(ImageEncoder):
    """[B,3,224,224] -> [B,256,196] (image_encoders.py:28-37)."""

    def __init__(self):
        super().__init__()
        self.base = nn.Sequential(*resnet18_children(upto_layer=3))

    def forward(self, x):
        x = self.base(x)
        return x.view(x.size(0), 256, -1)
