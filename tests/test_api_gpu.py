"""GPU: the reference's Python call surface exercised the way examples/main.py uses it -
backbones.load, NetworkFeatureAggregator, AnomalyClusteringCore.load/embed/_embed, PatchMaker,
Preprocessing / Aggregator modules, Matrix_Alpha_* - with return types and values checked against the oracle."""
import numpy as np
import pytest
import torch

from _synth import gaussish_tensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def core(dev):
    import patchcore.backbones
    import patchcore.common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    backbone = patchcore.backbones.load("dino_vitbase8")                 # seeded random init (no checkpoint offline)
    backbone.name, backbone.seed = "dino_vitbase8", None                 # main.py:235 sets these attributes
    assert next(backbone.parameters()).is_cuda and not any(p.requires_grad for p in backbone.parameters())
    c = AnomalyClusteringCore(dev)
    c.load(backbone=backbone, layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
           input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
           featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5,      # lands in **kwargs, as upstream
           nn_method=patchcore.common.FaissNN(True, 4))
    return c


def test_load_wires_the_reference_attributes(core):
    assert set(core.forward_modules.keys()) == {"feature_aggregator", "preprocessing", "preadapt_aggregator"}
    assert core.forward_modules["preprocessing"].output_dim == 2048
    assert core.forward_modules["preadapt_aggregator"].target_dim == 4096
    assert core.patch_maker.patchsize == 3 and core.layers_to_extract_from == ["blocks.10", "blocks.11"]
    assert len(core.backbone.hook_handles) == 2
    assert core.anomaly_scorer.n_nearest_neighbours == 1 and core.anomaly_segmentor.target_size == (224, 224)


def test_embed_dataloader_returns_reference_structure(core, dev):
    """embed(DataLoader of dict batches, batch size 1) -> (list per image of 784 numpy rows, list of is_anomaly),
    convertible with torch.tensor(...) exactly as examples/main.py:267 does."""
    from patchcore.datasets import synthetic
    from oracle import vit as ovit, aggregate as oagg
    ds = synthetic.SyntheticMVTecDataset(classname="toothbrush", split=synthetic.DatasetSplit.TEST, limit_per_class=1)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
    feats, labels = core.embed(loader, "unsupervised")
    assert len(feats) == 2 and len(feats[0]) == 784 and feats[0][0].shape == (4096,) and feats[0][0].dtype == np.float32
    assert [int(l) for l in labels] == [1, 0]
    Z = torch.tensor(np.array(feats))
    assert Z.shape == (2, 784, 4096)
    # the same two images through embed_device (batched) and through the oracle
    Zd = core.embed_device(ds.images, batch_size=2).cpu()
    np.testing.assert_allclose(Z.numpy(), Zd.numpy(), rtol=0, atol=5e-5)      # batch 1 vs batch 2: fp32-level
    sd = {k: v.cpu() for k, v in core.backbone.state_dict().items()}
    taps = ovit.forward_taps(sd, ds.images, taps=(10, 11))
    Zr = oagg.aggregate([taps[10], taps[11]], 3, 2048, 4096)[0].reshape(2, 784, 4096)
    err = (Zd - Zr).abs().max().item()
    print("Z max abs err vs oracle (12 blocks + aggregation): %.2e" % err)
    assert err <= 2e-4
    # tensor input, detach=False, provide_patch_shapes=True
    out, shapes = core._embed(ds.images[:1].to(dev), None, detach=False, provide_patch_shapes=True)
    assert isinstance(out, torch.Tensor) and out.shape == (784, 4096) and not out.requires_grad
    assert shapes == [[28, 28], [28, 28]]


def test_patch_maker_and_projection_modules_vs_oracle(core, dev):
    from oracle import aggregate as oagg
    x = torch.from_numpy(gaussish_tensor((2, 6, 9, 8), salt=95))
    patches, grid = core.patch_maker.patchify(x.to(dev), return_spatial_info=True)
    ref, rgrid = oagg.patchify(x, 3)
    assert grid == rgrid == [9, 8] and torch.equal(patches.cpu(), ref.contiguous())
    assert core.patch_maker.unpatch_scores(torch.zeros(6, 4), 2).shape == (2, 3, 4)
    assert float(core.patch_maker.score(np.arange(12.0).reshape(2, 3, 2))[1]) == 11.0
    # Preprocessing (MeanMapper per layer, stacked) and Aggregator as stand-alone modules
    import patchcore.common as common
    pa = torch.from_numpy(gaussish_tensor((5, 16, 3, 3), salt=96))
    pb = torch.from_numpy(gaussish_tensor((5, 24, 3, 3), salt=97))
    pre = common.Preprocessing([16, 24], 40)
    stacked = pre([pa.to(dev), pb.to(dev)])
    ref_stacked = torch.stack([oagg.mean_map(pa, 40), oagg.mean_map(pb, 40)], dim=1)
    np.testing.assert_allclose(stacked.cpu().numpy(), ref_stacked.numpy(), rtol=1e-6, atol=1e-7)
    agg = common.Aggregator(target_dim=50)
    ref_final = torch.nn.functional.adaptive_avg_pool1d(ref_stacked.reshape(5, 1, -1), 50).reshape(5, -1)
    np.testing.assert_allclose(agg(stacked).cpu().numpy(), ref_final.numpy(), rtol=1e-6, atol=1e-7)


def test_matrix_alpha_operators_signature_and_values(dev):
    from patchcore.utils import Matrix_Alpha_Unsupervised, Matrix_Alpha_Supervised, \
        Weight_Distance_Unsupervised, Weight_Distance_Supervised
    from oracle import alpha as oa
    from _synth import clustered_embeddings
    Zs, _ = clustered_embeddings(5, 64, 96, 2, salt=98)
    Zt, _ = clustered_embeddings(3, 64, 96, 1, salt=99)
    Z, Ztr = torch.from_numpy(Zs), torch.from_numpy(Zt)
    a = Matrix_Alpha_Unsupervised(tau=2, k=1, Z=Z.to(dev), device=dev)
    assert a.dtype == torch.float64 and a.shape == (5, 64) and a.is_cuda
    np.testing.assert_allclose(a.cpu().numpy(), oa.matrix_alpha_unsupervised(2, 1, Z).numpy(), rtol=2e-4, atol=1e-9)
    s = Matrix_Alpha_Supervised(tau=0.5, k=1, Z=Z.to(dev), Z_train=Ztr.to(dev), device=dev)
    np.testing.assert_allclose(s.cpu().numpy(), oa.matrix_alpha_supervised(0.5, 1, Z, Ztr).numpy(), rtol=4e-4, atol=1e-9)
    w2 = Weight_Distance_Unsupervised(Z.to(dev), 2, dev)
    np.testing.assert_allclose(w2.cpu().numpy(), oa.weight_distance_unsupervised(Z, 2).numpy(), rtol=0, atol=3e-5)
    w3 = Weight_Distance_Supervised(Z.to(dev), Ztr.to(dev), 3, dev)
    np.testing.assert_allclose(w3.cpu().numpy(), oa.weight_distance_supervised(Z, Ztr, 3).numpy(), rtol=0, atol=3e-5)


def test_final_norm_and_intermediate_layers(core, dev):
    """VisionTransformer.forward / get_intermediate_layers (off the hot path, but part of the module surface)."""
    from oracle import vit as ovit
    import torch.nn.functional as F
    model = core.backbone
    for h in model.hook_handles:            # plain forward: remove the early-exit hooks for this check
        h.remove()
    model.hook_handles = []
    images = torch.from_numpy(gaussish_tensor((1, 3, 224, 224), salt=100, scale=1.5))
    cls = model(images.to(dev)).cpu()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    x11 = ovit.forward_taps(sd, images, taps=(11,))[11]
    ref = F.layer_norm(x11, (768,), sd["norm.weight"], sd["norm.bias"], 1e-6)[:, 0]
    assert cls.shape == (1, 768)
    np.testing.assert_allclose(cls.numpy(), ref.numpy(), rtol=0, atol=3e-4)
    last = model.get_intermediate_layers(images.to(dev), n=1)[0].cpu()
    np.testing.assert_allclose(last.numpy(), F.layer_norm(x11, (768,), sd["norm.weight"], sd["norm.bias"], 1e-6).numpy(),
                               rtol=0, atol=3e-4)
