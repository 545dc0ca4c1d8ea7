"""GPU: BASELINE config 1 through the CLI modules - "screw"-shaped tiny subset (first 4 images per class, N = 24),
WideResNet50 backbone (torchvision's wide_resnet50_2 architecture on the MI355X, patchcore/wide_resnet.py; seeded
torchvision initialisation since the checkpoint is unreachable), layer2 [512,28,28] + layer3 [1024,14,14],
1024 -> 1024, patchsize 3, `average` weights: examples/main.py -> pickle -> examples/test.py against the oracle
(oracle/wrn.py taps -> oracle aggregation -> weighted average -> clustering) on the same images and weights.
Exercises the HIP convolution path, the CNN branch of the aggregation kernel (bilinear alignment of the 14x14 patch
grid to 28x28, channels-last maps read in place) and the artefact / CSV plumbing."""
import numpy as np
import pytest
import torch

from test_end_to_end_gpu import _load_examples

pytestmark = pytest.mark.gpu


def test_config1_screw_subset_average(tmp_path, dev, monkeypatch):
    main, test_mod = _load_examples()
    import patchcore.backbones
    import patchcore.common as common
    from patchcore.datasets import synthetic
    from oracle import aggregate as oagg, alpha as oalpha, cluster as ocluster

    orig = synthetic.SyntheticMVTecDataset
    monkeypatch.setattr(synthetic, "SyntheticMVTecDataset", lambda **kw: orig(limit_per_class=4, **kw))
    monkeypatch.setenv("AC_BACKBONE_SEED", "5")
    backbone = patchcore.backbones.load("wideresnet50")
    assert backbone.weights_source == "random-init(seed=5)"
    out = str(tmp_path / "outputs")
    layers = ["layer2", "layer3"]
    alpha, X = main.make_category_data(
        path="nowhere", category="screw", pretrain_embed_dimension=1024, target_embed_dimension=1024,
        backbone_names=["wideresnet50"], layers_to_extract_from=layers, patchsize=3,
        save_path=out + "/synthetic/wideresnet50/average", tau=1, train_ratio=1.0, supervised="average",
        dataset="synthetic", info_dir=out + "/synthetic/info", backbone=backbone, batch_size=8)
    assert alpha.shape == (24, 1, 784) and X.shape == (24, 1024)
    assert torch.allclose(alpha, torch.full_like(alpha, 1.0 / 784))

    ds = orig(classname="screw", split=synthetic.DatasetSplit.TEST, limit_per_class=4)
    assert len(ds) == 24
    from oracle import wrn as owrn
    sd = {k: v.detach().cpu() for k, v in backbone.state_dict().items()}
    feats = owrn.forward_taps(sd, ds.images)
    Z = oagg.aggregate([feats["layer2"], feats["layer3"]], 3, 1024, 1024)[0].reshape(24, 784, 1024)
    _, X_ref = oalpha.weighted_average(oalpha.matrix_alpha_average(Z), Z)
    print("config 1: X max abs err %.2e (|X| max %.2f)" % (np.abs(X - X_ref).max(), np.abs(X_ref).max()))
    np.testing.assert_allclose(X, X_ref, rtol=0, atol=1e-4)

    NMI, ARI, F1, label, predict = test_mod.calculate_metrics(
        category="screw", pretrain_embed_dimension=1024, target_embed_dimension=1024, backbone_names=["wideresnet50"],
        layers_to_extract_from=layers, patchsize=3, train_ratio=1.0, tau=1, supervised="average", dataset="synthetic",
        output_dir=out)
    rN, rA, rF, rl, rp = ocluster.calculate_metrics(X_ref, list(ds.anomalies))
    print("config 1: NMI/ARI/F1 ours %.4f %.4f %.4f  oracle %.4f %.4f %.4f" % (NMI, ARI, F1, rN, rA, rF))
    assert abs(NMI - rN) <= 1e-3 and abs(ARI - rA) <= 1e-3 and abs(F1 - rF) <= 1e-3
    assert np.array_equal(label, rl) and len(set(label)) == 6
