"""GPU: wide_resnet50_2 on the MI355X (convolutions as split-fp16 MFMA GEMMs with fused BatchNorm / residual / ReLU,
csrc/cnn.hip) against the torch-CPU restatement of torchvision's network (oracle/wrn.py) on the same weights and
images - layer2 / layer3 taps through the hook mechanism, then through the aggregation kernel's channels-last path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trained_like(model, seed=7):
    """Non-trivial BatchNorm statistics (the init has mean 0 / var 1 / weight 1 / bias 0, which would hide a wrong
    fold): running_var in [0.5, 2], running_mean and bias ~ +-0.3, weight in [0.7, 1.3]."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                c = m.weight.shape[0]
                m.running_var.copy_(0.5 + 1.5 * torch.rand(c, generator=g))
                m.running_mean.copy_(0.6 * torch.rand(c, generator=g) - 0.3)
                m.bias.copy_(0.6 * torch.rand(c, generator=g) - 0.3)
                m.weight.copy_(0.7 + 0.6 * torch.rand(c, generator=g))
    return model


@pytest.fixture(scope="module")
def wrn(dev):
    from patchcore import wide_resnet
    torch.manual_seed(0)
    m = _trained_like(wide_resnet.wide_resnet50_2()).eval()
    for p in m.parameters():
        p.requires_grad = False
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to(dev), sd


def _close(got, ref, what):
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    print("%s: max |err| %.2e at |x| <= %.2f (relative to the largest value %.1e)" % (what, err, scale, err / scale))
    assert err <= 1e-5 * scale + 1e-6, what


def test_layer_taps_match_the_cpu_restatement(dev, wrn):
    import patchcore.common as common
    from oracle import wrn as owrn
    model, sd = wrn
    g = torch.Generator().manual_seed(11)
    images = torch.randn((3, 3, 224, 224), generator=g)
    agg = common.NetworkFeatureAggregator(model, ["layer2", "layer3"], dev)
    out = dict(agg(images.to(dev)))          # the aggregator hands out its own dict (as upstream): keep this call's taps
    ref = owrn.forward_taps(sd, images)
    assert tuple(out["layer2"].shape) == (3, 512, 28, 28) and tuple(out["layer3"].shape) == (3, 1024, 14, 14)
    assert agg.feature_dimensions((3, 224, 224)) == [512, 1024]
    for name in ("layer2", "layer3"):
        _close(out[name].cpu(), ref[name], name)
    # an earlier tap stops earlier (LastLayerToExtractReachedException semantics), odd sizes go through the clamps
    agg1 = common.NetworkFeatureAggregator(model, ["layer1"], dev)
    small = torch.randn((2, 3, 72, 104), generator=g)
    o1 = agg1(small.to(dev))
    assert list(o1) == ["layer1"]
    _close(o1["layer1"].cpu(), owrn.forward_taps(sd, small, taps=("layer1",))["layer1"], "layer1 at 72x104")


def test_embed_through_the_channels_last_aggregation(dev, wrn):
    """AnomalyClusteringCore on the WRN taps (1024 -> 1024, patch size 3: config 1's dimensions): the aggregation
    kernel reads the channels-last maps in place; Z against the oracle aggregation of the oracle taps."""
    import patchcore.common as common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from oracle import wrn as owrn, aggregate as oagg
    model, sd = wrn
    core = AnomalyClusteringCore(dev)
    core.load(backbone=model, layers_to_extract_from=["layer2", "layer3"], device=dev, input_shape=(3, 224, 224),
              pretrain_embed_dimension=1024, target_embed_dimension=1024, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), nn_method=common.FaissNN(False, 4))
    g = torch.Generator().manual_seed(12)
    images = torch.randn((2, 3, 224, 224), generator=g)
    Z = core.embed_device(images, batch_size=2)
    ref = owrn.forward_taps(sd, images)
    Zr = oagg.aggregate([ref["layer2"], ref["layer3"]], 3, 1024, 1024)[0].reshape(2, 784, 1024)
    assert tuple(Z.shape) == (2, 784, 1024)
    err = float((Z.cpu() - Zr).abs().max())
    print("Z (standardised per image): max |err| %.2e" % err)
    assert err <= 2e-5


def test_nan_pixels_propagate_like_torch(dev, wrn):
    """Round-3 advice: fmaxf() drops a NaN, so the max pool and the ReLU epilogues used to turn a poisoned pixel into a
    finite, meaningless feature; nn.MaxPool2d / nn.ReLU (what the reference's torchvision network runs) keep it, and
    it then reaches cdist / pdist where scipy raises.  A NaN in the input image must poison the same taps as on the
    CPU restatement (which uses torch's own ops), and leave the other image of the batch untouched."""
    import patchcore.common as common
    from oracle import wrn as owrn
    model, sd = wrn
    g = torch.Generator().manual_seed(13)
    images = torch.randn((2, 3, 64, 64), generator=g)
    images[0, 1, 20, 21] = float("nan")
    agg = common.NetworkFeatureAggregator(model, ["layer1", "layer2"], dev)
    out = dict(agg(images.to(dev)))
    ref = owrn.forward_taps(sd, images, taps=("layer1", "layer2"))
    for name in ("layer1", "layer2"):
        mine, r = out[name].cpu(), ref[name]
        assert torch.equal(torch.isnan(mine), torch.isnan(r)), name
        assert torch.isnan(mine[0]).any() and not torch.isnan(mine[1]).any()
        ok = ~torch.isnan(r)
        assert float((mine[ok] - r[ok]).abs().max()) <= 1e-5 * float(r[ok].abs().max()) + 1e-6
