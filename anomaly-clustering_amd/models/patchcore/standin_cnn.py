"""Seeded WRN50-SHAPED stand-in CNN for BASELINE config 1 (CPU/CLI plumbing) when torchvision is absent.

The reference's ``wideresnet50`` entry is ``models.wide_resnet50_2(pretrained=True)`` (backbones.py:27) tapped at
``layer2`` / ``layer3`` (examples/main.py defaults for CNN backbones): feature maps [B,512,28,28] and
[B,1024,14,14] at 224x224 input.  Neither torchvision nor its checkpoint is reachable here, so this module
reproduces only the SHAPES and the module names the hook mechanism resolves (common.py:238-247): a strided
convolution per stage, ReLU, Kaiming-normal weights from a torch.Generator seeded by the caller.

OFF the MI355X hot path: plain torch convolutions (MIOpen on the GPU, or CPU).  What it exercises is the
CNN branch of the aggregation kernel (bilinear grid alignment 14x14 -> 28x28, 1024 -> 1024) and the
main.py -> test.py plumbing of config 1.
"""
import torch
from torch import nn


class _Stage(nn.Sequential):
    def __init__(self, cin, cout, stride):
        super().__init__(nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False), nn.ReLU(inplace=False))


class WRN50Shaped(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)        # 224 -> 112
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)                      # -> 56
        self.layer1 = _Stage(64, 256, 1)                                         # [256, 56, 56]
        self.layer2 = _Stage(256, 512, 2)                                        # [512, 28, 28]
        self.layer3 = _Stage(512, 1024, 2)                                       # [1024, 14, 14]
        self.layer4 = _Stage(1024, 2048, 2)                                      # never reached on the path

    def forward(self, x):
        x = self.maxpool(self.relu(self.conv1(x)))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def wrn50_shaped(seed=0):
    model = WRN50Shaped()
    g = torch.Generator().manual_seed(int(seed))
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            fan_out = m.weight.shape[0] * m.weight.shape[2] * m.weight.shape[3]
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
    for p in model.parameters():
        p.requires_grad = False
    return model.eval()
