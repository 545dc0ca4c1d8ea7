"""CPU: the wide_resnet50_2 module (reference backbones.py:27 -> torchvision, absent here) without a GPU:
torchvision's state_dict key list and shapes, strict loading of a checkpoint in that format, the layer names the
hook mechanism resolves, the oracle restatement's tap shapes, and the refusal to run on the CPU."""
import pytest
import torch


def test_state_dict_is_torchvisions():
    from oracle import wrn as owrn
    from patchcore import wide_resnet
    torch.manual_seed(0)
    m = wide_resnet.wide_resnet50_2()
    sd = m.state_dict()
    want = owrn.state_dict_shapes()
    assert list(sd.keys()) == list(want.keys())                 # torchvision's names, in torchvision's order
    assert all(tuple(sd[k].shape) == want[k] for k in want)
    assert sum(v.numel() for k, v in sd.items() if "num_batches" not in k and "running" not in k) == 68883240
    # torchvision's initialisation: kaiming-normal(fan_out) convolutions, BatchNorm weight 1 / bias 0
    w = sd["layer2.0.conv2.weight"]
    assert abs(float(w.std()) - (2.0 / (w.shape[0] * 9)) ** 0.5) < 0.02 * (2.0 / (w.shape[0] * 9)) ** 0.5
    assert torch.all(sd["layer3.5.bn3.weight"] == 1) and torch.all(sd["bn1.bias"] == 0)
    torch.manual_seed(0)
    m2 = wide_resnet.wide_resnet50_2()
    assert all(torch.equal(a, b) for a, b in zip(sd.values(), m2.state_dict().values()))   # seeded -> reproducible


def test_checkpoint_loading_and_hook_names(tmp_path, monkeypatch):
    import patchcore.backbones as bb
    import patchcore.common as common
    from patchcore import wide_resnet
    torch.manual_seed(1)
    src = wide_resnet.wide_resnet50_2()
    with torch.no_grad():
        src.bn1.running_mean.add_(0.25)
    torch.save(src.state_dict(), tmp_path / "wide_resnet50_2-95faca4d.pth")
    monkeypatch.delenv("AC_BACKBONE_SEED", raising=False)
    with pytest.raises(RuntimeError, match="AC_WRN50_WEIGHTS"):
        bb.load("wideresnet50")
    monkeypatch.setenv("AC_WRN50_WEIGHTS", str(tmp_path))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    m = bb.load("wideresnet50")
    assert m.weights_source.endswith("wide_resnet50_2-95faca4d.pth")
    assert torch.equal(m.state_dict()["bn1.running_mean"], src.state_dict()["bn1.running_mean"])
    assert common._resolve(m, "layer2") is m.layer2 and common._resolve(m, "layer3.5") is m.layer3[5]
    bad = dict(src.state_dict())
    bad.pop("layer1.0.conv1.weight")
    torch.save(bad, tmp_path / "wide_resnet50_2-95faca4d.pth")
    with pytest.raises(RuntimeError):
        bb.load("wideresnet50")                                  # strict=True
    monkeypatch.setenv("AC_WRN50_WEIGHTS", str(tmp_path / "nowhere"))
    with pytest.raises(FileNotFoundError):
        bb.load("wideresnet50")


def test_oracle_taps_and_no_cpu_forward():
    from oracle import wrn as owrn
    from patchcore import wide_resnet
    torch.manual_seed(2)
    m = wide_resnet.wide_resnet50_2().eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    taps = owrn.forward_taps(sd, torch.randn(2, 3, 64, 64))
    assert tuple(taps["layer2"].shape) == (2, 512, 8, 8) and tuple(taps["layer3"].shape) == (2, 1024, 4, 4)
    assert list(owrn.forward_taps(sd, torch.randn(1, 3, 32, 32), taps=("layer1",))) == ["layer1"]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
