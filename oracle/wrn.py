"""Oracle: torchvision's ``wide_resnet50_2`` forward up to ``layer3``, on the CPU.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

The reference's ``wideresnet50`` backbone is ``models.wide_resnet50_2(pretrained=True)`` (models/patchcore/
backbones.py:27), tapped at ``layer2`` / ``layer3`` by forward hooks with an early exit after the last one
(models/patchcore/common.py:211-292; README "WideResNet50" rows).  The arithmetic lives in a THIRD-PARTY
dependency that is absent from /root/reference and from this image: torchvision (requirements.txt pins
``torchvision==0.13.1``; ``torchvision/models/resnet.py``: ``ResNet(Bottleneck, [3, 4, 6, 3],
width_per_group=64 * 2)``).  This file restates that published architecture as a function of a torchvision-format
``state_dict``:

    conv1 7x7/2 pad 3 (3 -> 64, no bias) - bn1 - relu - maxpool 3x3/2 pad 1
    layerL, L = 1..4: blocks (3, 4, 6, 3) of Bottleneck(planes = 64 * 2^(L-1)), expansion 4, width = 2 * planes:
        conv1 1x1 (in -> width) - bn1 - relu - conv2 3x3 pad 1 (stride s ON THE 3x3: "ResNet v1.5") - bn2 - relu -
        conv3 1x1 (width -> 4 * planes) - bn3 - (+ identity, through downsample = conv 1x1 stride s + bn when the
        shape changes) - relu;  s = 2 for the first block of layers 2-4, else 1
    BatchNorm2d in eval mode: (x - running_mean) / sqrt(running_var + 1e-5) * weight + bias.

PARITY UNPINNED: torchvision cannot be imported here, so no output of the reference's own network exists to check
this restatement against; it is anchored on the published module definition above and on the state_dict key
list / tensor shapes of torchvision's checkpoint (tests/test_wrn_cpu.py).
"""
import torch
import torch.nn.functional as F

BLOCKS = (3, 4, 6, 3)
EPS = 1e-5


def _bn(sd, pre, x):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        training=False, eps=EPS)


def bottleneck(sd, pre, x, stride):
    out = F.relu(_bn(sd, pre + "bn1.", F.conv2d(x, sd[pre + "conv1.weight"])))
    out = F.relu(_bn(sd, pre + "bn2.", F.conv2d(out, sd[pre + "conv2.weight"], stride=stride, padding=1)))
    out = _bn(sd, pre + "bn3.", F.conv2d(out, sd[pre + "conv3.weight"]))
    if pre + "downsample.0.weight" in sd:
        x = _bn(sd, pre + "downsample.1.", F.conv2d(x, sd[pre + "downsample.0.weight"], stride=stride))
    return F.relu(out + x)


@torch.no_grad()
def forward_taps(sd, images, taps=("layer2", "layer3")):
    """images fp32 [B, 3, H, W] -> {layer name: [B, C, h, w]} for the requested layers; stops after the last one
    (LastLayerToExtractReachedException semantics, common.py:259-268)."""
    x = F.relu(_bn(sd, "bn1.", F.conv2d(images, sd["conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    out = {}
    last = max(int(t[-1]) for t in taps)
    for L in range(1, last + 1):
        for b in range(BLOCKS[L - 1]):
            x = bottleneck(sd, "layer%d.%d." % (L, b), x, 2 if (b == 0 and L > 1) else 1)
        name = "layer%d" % L
        if name in taps:
            out[name] = x
    return out


def state_dict_shapes():
    """Key -> shape of torchvision's wide_resnet50_2 state_dict (what ``$AC_WRN50_WEIGHTS`` must hold)."""
    shapes = {"conv1.weight": (64, 3, 7, 7)}

    def bn(pre, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            shapes[pre + k] = (c,)
        shapes[pre + "num_batches_tracked"] = ()

    bn("bn1.", 64)
    inplanes = 64
    for L in range(1, 5):
        planes = 64 * 2 ** (L - 1)
        width, outp = 2 * planes, 4 * planes
        for b in range(BLOCKS[L - 1]):
            pre = "layer%d.%d." % (L, b)
            shapes[pre + "conv1.weight"] = (width, inplanes, 1, 1)
            bn(pre + "bn1.", width)
            shapes[pre + "conv2.weight"] = (width, width, 3, 3)
            bn(pre + "bn2.", width)
            shapes[pre + "conv3.weight"] = (outp, width, 1, 1)
            bn(pre + "bn3.", outp)
            if b == 0:
                shapes[pre + "downsample.0.weight"] = (outp, inplanes, 1, 1)
                bn(pre + "downsample.1.", outp)
            inplanes = outp
    shapes["fc.weight"] = (1000, 2048)
    shapes["fc.bias"] = (1000,)
    return shapes
