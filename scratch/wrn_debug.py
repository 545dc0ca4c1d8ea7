"""Stage-by-stage comparison of the HIP wide_resnet50_2 layers with torch-CPU convolutions (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "anomaly-clustering_amd", "models"), os.path.join(ROOT, "tests")]
import torch
import torch.nn.functional as F
from patchcore import wide_resnet
from patchcore.wide_resnet import _ConvBN, _nhwc
import acnative
from test_wrn_gpu import _trained_like

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = _trained_like(wide_resnet.wide_resnet50_2()).eval()
cpu = {k: v.detach().clone() for k, v in m.state_dict().items()}
m = m.to(dev)
g = torch.Generator().manual_seed(11)
x = torch.randn((2, 3, 224, 224), generator=g)


def bn(pre, t):
    return F.batch_norm(t, cpu[pre + "running_mean"], cpu[pre + "running_var"], cpu[pre + "weight"], cpu[pre + "bias"], False, eps=1e-5)


def cmp(name, got_nhwc, ref_nchw):
    got = got_nhwc.permute(0, 3, 1, 2).cpu()
    print("%-28s shape %s err %.3e scale %.3e" % (name, tuple(got.shape), float((got - ref_nchw).abs().max()), float(ref_nchw.abs().max())))


stem = _ConvBN(m.conv1, m.bn1)
y = stem(x.to(dev).permute(0, 2, 3, 1))
r = F.relu(bn("bn1.", F.conv2d(x, cpu["conv1.weight"], stride=2, padding=3)))
cmp("stem conv7x7+bn+relu", y, r)
y = acnative.maxpool_nhwc(y, 3, 2, 1)
r = F.max_pool2d(r, 3, 2, 1)
cmp("maxpool", y, r)
blk = m.layer1[0]
pk = blk.packed()
# feed the REFERENCE activations into each HIP conv so errors do not compound
rin = r
xin = rin.permute(0, 2, 3, 1).contiguous().to(dev)
y1 = pk["c1"](xin)
r1 = F.relu(bn("layer1.0.bn1.", F.conv2d(rin, cpu["layer1.0.conv1.weight"])))
cmp("layer1.0 conv1 1x1", y1, r1)
y2 = pk["c2"](r1.permute(0, 2, 3, 1).contiguous().to(dev))
r2 = F.relu(bn("layer1.0.bn2.", F.conv2d(r1, cpu["layer1.0.conv2.weight"], padding=1)))
cmp("layer1.0 conv2 3x3", y2, r2)
yd = pk["ds"](xin, relu=False)
rd = bn("layer1.0.downsample.1.", F.conv2d(rin, cpu["layer1.0.downsample.0.weight"]))
cmp("layer1.0 downsample", yd, rd)
y3 = pk["c3"](r2.permute(0, 2, 3, 1).contiguous().to(dev), res=rd.permute(0, 2, 3, 1).contiguous().to(dev), relu=True)
r3 = F.relu(bn("layer1.0.bn3.", F.conv2d(r2, cpu["layer1.0.conv3.weight"])) + rd)
cmp("layer1.0 conv3+res+relu", y3, r3)
# stride-2 block
blk = m.layer2[0]
pk = blk.packed()
g2 = torch.Generator().manual_seed(5)
rin = torch.randn((2, 256, 56, 56), generator=g2)
xin = rin.permute(0, 2, 3, 1).contiguous().to(dev)
r1 = F.relu(bn("layer2.0.bn1.", F.conv2d(rin, cpu["layer2.0.conv1.weight"])))
cmp("layer2.0 conv1", pk["c1"](xin), r1)
r2 = F.relu(bn("layer2.0.bn2.", F.conv2d(r1, cpu["layer2.0.conv2.weight"], stride=2, padding=1)))
cmp("layer2.0 conv2 3x3/2", pk["c2"](r1.permute(0, 2, 3, 1).contiguous().to(dev)), r2)
rd = bn("layer2.0.downsample.1.", F.conv2d(rin, cpu["layer2.0.downsample.0.weight"], stride=2))
cmp("layer2.0 downsample 1x1/2", pk["ds"](xin, relu=False), rd)
out = blk(rin.to(dev).contiguous(memory_format=torch.channels_last))
from oracle import wrn as owrn
cmp("layer2.0 whole block", out.permute(0, 2, 3, 1), owrn.bottleneck(cpu, "layer2.0.", rin, 2))

print("---- every block on the oracle's input, batch 3")
x3 = torch.randn((3, 3, 224, 224), generator=g)
r = F.max_pool2d(F.relu(bn("bn1.", F.conv2d(x3, cpu["conv1.weight"], stride=2, padding=3))), 3, 2, 1)
for L in (1, 2, 3):
    layer = getattr(m, "layer%d" % L)
    for b, blk in enumerate(layer):
        ref = owrn.bottleneck(cpu, "layer%d.%d." % (L, b), r, 2 if (b == 0 and L > 1) else 1)
        got = blk(r.to(dev).contiguous(memory_format=torch.channels_last))
        cmp("layer%d.%d" % (L, b), got.permute(0, 2, 3, 1), ref)
        r = ref
print("---- chained")
y = m._stem(x3.to(dev).permute(0, 2, 3, 1)) if m._stem is not None else None
if y is None:
    m._stem = _ConvBN(m.conv1, m.bn1)
    y = m._stem(x3.to(dev).permute(0, 2, 3, 1))
y = acnative.maxpool_nhwc(y, 3, 2, 1).permute(0, 3, 1, 2)
r = F.max_pool2d(F.relu(bn("bn1.", F.conv2d(x3, cpu["conv1.weight"], stride=2, padding=3))), 3, 2, 1)
for L in (1, 2, 3):
    layer = getattr(m, "layer%d" % L)
    for b, blk in enumerate(layer):
        r = owrn.bottleneck(cpu, "layer%d.%d." % (L, b), r, 2 if (b == 0 and L > 1) else 1)
        y = blk(y)
        cmp("chained layer%d.%d" % (L, b), y.permute(0, 2, 3, 1), r)
