"""GPU: BASELINE config 2 at its REAL size - a whole MVTec-shaped category (N = 83 "bottle", N = 167 "pill")
through all 12 ViT blocks, aggregation and the alpha engine - checked through properties that need no CPU
oracle run at that size:

  * alpha rows sum to 1 (fp64 softmax rounded to fp32: |sum - 1| <= 1e-5), finite, positive;
  * the symmetric triangle-grid path (65 072 / 130 928 patch rows - far beyond the 3 920 rows of the small parity
    cases) equals the general path (distinct key buffer: full rectangle, one-way) to 1e-5 = 5 ulp of fp32 at
    w ~ 30 (the two paths form a.b and b.a, whose three split-fp16 passes accumulate in a different order;
    measured 3.8e-6 = 2 ulp);
  * w of three sampled images equals the float64 brute force (fp64 GEMM on the GPU via torch - test-side
    arithmetic) to 2e-5: the tolerance of tests/test_alpha_gpu.py at full size, i.e. the distance of the
    reference's own fp32 cdist from the truth;
  * supervised at N = 83 / M = 83 (bank = 83 seeded train images): same check against the fp64 brute force.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W_TOL = 2e-5


@pytest.fixture(scope="module")
def core(dev):
    import patchcore.backbones
    import patchcore.common as common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    backbone = patchcore.backbones.load("dino_vitbase8")          # seeded init (AC_BACKBONE_SEED=0, conftest.py)
    c = AnomalyClusteringCore(dev)
    c.load(backbone=backbone, layers_to_extract_from=["blocks.10", "blocks.11"], device=dev, input_shape=(3, 224, 224),
           pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
           featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    return c


def _truth_rows(Zq, Zk, idx, reduce, self_offset=0):
    """float64 w of the query images idx (torch fp64 on the GPU)."""
    Zk64 = Zk.double()
    ksq = (Zk64 * Zk64).sum(2)
    out = []
    for i in idx:
        q = Zq[i].double()
        qsq = (q * q).sum(1)
        cols = []
        for j in range(Zk.shape[0]):
            if reduce == 0 and j == i + self_offset:
                continue
            d2 = qsq[:, None] + ksq[j][None, :] - 2.0 * (q @ Zk64[j].T)
            cols.append(d2.clamp_min(0).sqrt().min(1)[0])
        cols = torch.stack(cols, 1)
        out.append(cols.mean(1) if reduce == 0 else cols.min(1)[0])
    return torch.stack(out)


@pytest.mark.parametrize("category,n", [("bottle", 83), ("pill", 167)])
def test_unsupervised_category_at_full_size(dev, core, category, n):
    import acnative
    from patchcore.pipeline import HipOps, run_category
    from patchcore.datasets import synthetic
    labels = synthetic.category_labels(category)
    assert len(labels) == n
    images = synthetic.make_images(category, labels).to(dev)
    ops = HipOps(core, batch_size=128)
    out = run_category(ops, images, "unsupervised", tau=1.0, want_host=False)
    alpha = out["alpha"].squeeze(1)
    assert alpha.shape == (n, 784) and torch.isfinite(alpha).all() and (alpha > 0).all()
    assert (alpha.double().sum(1) - 1.0).abs().max().item() <= 1e-5
    assert out["X_dev"].shape == (n, 4096) and torch.isfinite(out["X_dev"]).all()
    assert out["pdist"].shape == (n * (n - 1) // 2,) and torch.isfinite(out["pdist"]).all()

    Z = ops.embed(images)
    w_sym = out["w_loc"]
    assert torch.equal(w_sym, acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS))     # reproducible
    w_gen = acnative.pair_min_dist_staged(Z, Z.clone(), acnative.REDUCE_MEAN_OTHERS)
    assert (w_sym - w_gen).abs().max().item() <= 1e-5, "symmetric triangle grid vs general rectangle"
    idx = [0, n // 2, n - 1]
    truth = _truth_rows(Z, Z, idx, 0)
    err = (w_sym[idx].double() - truth).abs().max().item()
    print("%s N=%d: |w - fp64 truth| = %.2e (w ~ %.1f)" % (category, n, err, truth.mean().item()))
    assert err <= W_TOL
    # the sub-blocked ring schedule (what N = 10 000 runs) on the same Z: same bits
    blocked = run_category(ops, images, "unsupervised", tau=1.0, want_host=False, budget=40 * 40 * 784 * 4)
    assert torch.equal(blocked["w_loc"], w_sym)


def test_supervised_category_at_full_size(dev, core):
    """N = 83 test images against a bank of M = 83 train images (config 4's shape for one category)."""
    from patchcore.pipeline import HipOps, run_category, select_bank
    from patchcore.datasets import synthetic
    labels = synthetic.category_labels("bottle")
    images = synthetic.make_images("bottle", labels).to(dev)
    ntr = synthetic.TRAIN_SIZES["bottle"]
    train = synthetic.make_images("bottle", ["good"] * ntr, split=synthetic.DatasetSplit.TRAIN).to(dev)
    ops = HipOps(core, batch_size=128)
    out = run_category(ops, images, "supervised", tau=1.0, train_images=train, train_ratio=1.0, want_host=False)
    alpha = out["alpha"].squeeze(1)
    assert (alpha.double().sum(1) - 1.0).abs().max().item() <= 1e-5 and torch.isfinite(alpha).all()
    sel = select_bank(ntr, 83, 1.0, 0)
    assert len(sel) == 83
    Z, bank = ops.embed(images), ops.embed(train[sel.to(dev)])
    idx = [0, 41, 82]
    truth = _truth_rows(Z, bank, idx, 1)
    err = (out["w_loc"][idx].double() - truth).abs().max().item()
    print("supervised N=83 M=83: |w - fp64 truth| = %.2e" % err)
    assert err <= W_TOL
