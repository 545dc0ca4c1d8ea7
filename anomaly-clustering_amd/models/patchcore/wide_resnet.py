"""torchvision's ``wide_resnet50_2`` - the reference's ``wideresnet50`` backbone (models/patchcore/backbones.py:27:
``models.wide_resnet50_2(pretrained=True)``) - with its forward on the MI355X: every convolution is a GEMM on the
split-fp16 MFMA engine with BatchNorm / residual / ReLU fused into the epilogue (csrc/cnn.hip).

torchvision is not importable here, so this module restates its published definition (``torchvision/models/
resnet.py``: ``ResNet(Bottleneck, [3, 4, 6, 3], width_per_group=128)``, stride on the 3x3 convolution) with the
SAME module names, creation order and initialisation, so that

  * ``load_state_dict(torch.load("wide_resnet50_2-*.pth"), strict=True)`` accepts torchvision's checkpoint
    (``$AC_WRN50_WEIGHTS``; key list / shapes: tests/test_wrn_cpu.py), and
  * the hook mechanism of models/patchcore/common.py:211-292 resolves ``layer2`` / ``layer3`` (and ``layerN.i``)
    exactly as on torchvision's model: a layer is an ``nn.Sequential`` of ``Bottleneck`` modules, the hook sits
    on the last block and receives its output.

Activations travel between the blocks as ``[B, C, H, W]`` tensors in CHANNELS-LAST memory (a permuted view of the
NHWC rows the GEMM epilogue writes): what a hook sees has the reference's shape, and the aggregation kernel reads
it in place (``AC_FEAT_NHWC``).  The stem (conv1 - bn1 - relu - maxpool) runs fused, so hooks on those four
modules do not fire; ``layer4`` / ``avgpool`` / ``fc`` hold their parameters (strict checkpoint loading) but the
clustering path never reaches them - calling them raises.
"""
import torch
from torch import nn

import acnative

BLOCKS = (3, 4, 6, 3)


def _bn_fold(bn):
    """eval-mode BatchNorm as torch's CPU kernel evaluates it: y = x * alpha + beta."""
    invstd = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    alpha = (invstd * bn.weight.detach().float()).contiguous()
    beta = (bn.bias.detach().float() - bn.running_mean.detach().float() * alpha).contiguous()
    return alpha, beta


def _pack_conv(conv):
    """[Cout, Cin, kh, kw] -> hl32 [Cout, Kpad], k = (ky*kw + kx)*Cin + c, zero padded to a multiple of 32."""
    w = conv.weight.detach().float().permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1)
    K = w.shape[1]
    Kpad = (K + 31) // 32 * 32
    if Kpad != K:
        w = torch.cat([w, torch.zeros(w.shape[0], Kpad - K, dtype=w.dtype, device=w.device)], dim=1)
    return acnative.pack_hl32(w.contiguous()), Kpad


def _nhwc(x):
    """[B, C, H, W] channels-last tensor -> its [B, H, W, C] contiguous view."""
    v = x.permute(0, 2, 3, 1)
    if x.dtype != torch.float32 or not v.is_contiguous():
        raise ValueError("the HIP Bottleneck expects the fp32 channels-last activation the preceding layer wrote "
                         "(shape %s, strides %s)" % (tuple(x.shape), x.stride()))
    return v


class _ConvBN:
    """Device-side image of a (Conv2d, BatchNorm2d) pair, built on first use."""

    def __init__(self, conv, bn):
        self.w, self.K = _pack_conv(conv)
        self.alpha, self.beta = _bn_fold(bn)
        self.cout = conv.weight.shape[0]
        self.k, self.stride, self.pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]

    def __call__(self, x_nhwc, res=None, relu=True):
        """x fp32 [B, H, W, C] (any strides for the image stem) -> fp32 [B, Ho, Wo, Cout]."""
        B, H, W, C = x_nhwc.shape
        Ho = (H + 2 * self.pad - self.k) // self.stride + 1
        Wo = (W + 2 * self.pad - self.k) // self.stride + 1
        if self.k == 1 and self.stride == 1 and x_nhwc.is_contiguous():
            A = acnative.pack_hl32(x_nhwc.reshape(B * H * W, C))
        else:
            A = acnative.im2col_hl32(x_nhwc, self.k, self.k, self.stride, self.pad, self.K)
        out = acnative.conv_gemm_bn(A, self.w, B * Ho * Wo, self.cout, self.K, self.alpha, self.beta,
                                    None if res is None else res.reshape(B * Ho * Wo, self.cout), relu)
        return out.view(B, Ho, Wo, self.cout)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, base_width=128):
        super().__init__()
        width = int(planes * (base_width / 64.0))
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self._dev = None

    def packed(self):
        if self._dev is None:
            self._dev = {"c1": _ConvBN(self.conv1, self.bn1), "c2": _ConvBN(self.conv2, self.bn2),
                         "c3": _ConvBN(self.conv3, self.bn3),
                         "ds": _ConvBN(self.downsample[0], self.downsample[1]) if self.downsample is not None else None}
        return self._dev

    def forward(self, x):
        pk = self.packed()
        xn = _nhwc(x)
        out = pk["c2"](pk["c1"](xn))
        identity = xn if pk["ds"] is None else pk["ds"](xn, relu=False)
        out = pk["c3"](out, res=identity.contiguous(), relu=True)
        return out.permute(0, 3, 1, 2)


class WideResNet50_2(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, BLOCKS[0], 1)
        self.layer2 = self._make_layer(128, BLOCKS[1], 2)
        self.layer3 = self._make_layer(256, BLOCKS[2], 2)
        self.layer4 = self._make_layer(512, BLOCKS[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * Bottleneck.expansion, 1000)
        for m in self.modules():                        # torchvision's initialisation, in its order
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._stem = None

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._stem = None
        for m in self.modules():
            if isinstance(m, Bottleneck):
                m._dev = None
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)                 # .to(device): the device images are rebuilt on next use
        self._stem = None
        for m in self.modules():
            if isinstance(m, Bottleneck):
                m._dev = None
        return out

    def forward(self, x):
        if self._stem is None:
            self._stem = _ConvBN(self.conv1, self.bn1)
        x = x.to(torch.float32)
        y = self._stem(x.permute(0, 2, 3, 1))              # the NCHW image through its strides: no transposing copy
        y = acnative.maxpool_nhwc(y, 3, 2, 1).permute(0, 3, 1, 2)
        y = self.layer1(y)
        y = self.layer2(y)
        y = self.layer3(y)
        raise NotImplementedError("wide_resnet50_2 beyond layer3 is outside the anomaly-clustering path (the reference "
                                  "stops at the last tapped layer: common.py:259-268); tap layer1..layer3")


def wide_resnet50_2():
    return WideResNet50_2()
