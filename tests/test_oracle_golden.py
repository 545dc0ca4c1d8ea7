"""CPU: pin the oracle (oracle/*.py) to the reference.

Every comparison here is against vectors that tests/golden/make_golden.py captured by
running the REFERENCE's own code in the build container (ViT, aggregation, alpha), or
against the reference's committed artefacts (known-answer clustering rows).
"""
import os

import numpy as np
import pytest
import torch

from _synth import clustered_embeddings, gaussish_tensor, dino_like_statistics


# --------------------------------------------------------------------- helpers
def _seed0_vit(depth=12):
    """The product's ViT module under torch.manual_seed(0) (CPU tensors; never run forward here)."""
    import patchcore.vision_transformer as vits
    torch.manual_seed(0)
    m = vits.vit_base(patch_size=8, num_classes=0, depth=depth)
    return m


# ------------------------------------------------------------------------ ViT
def test_oracle_vit_with_dino_like_statistics(golden):
    """The reference's vit_base on weights with outlier channels, LN gains in [0.2, 5], non-zero biases and
    heavy-tailed matrices (tests/_synth.py::dino_like_statistics; golden written by make_golden.py --dino):
    the oracle restatement reproduces its block-10/11 outputs (|x| up to 112) to fp32 round-off."""
    from oracle import vit as ovit
    g = golden("vit_forward_dino_stats.npz")
    sd = dino_like_statistics(_seed0_vit().state_dict())
    names = sorted(sd.keys())
    for name, chk in zip(names, g["weight_checks"]):           # the recipe rebuilds the same weights
        t = sd[name].double().flatten()
        np.testing.assert_array_equal([t.sum().item(), (t * t).sum().item(), t[0].item(), t[-1].item()], chk, err_msg=name)
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=11, scale=1.7))
    taps = ovit.forward_taps(sd, images, taps=(10, 11))
    toks = g["tokens"]
    for tap, key in ((10, "f10"), (11, "f11")):
        ref = g[key]
        np.testing.assert_allclose(taps[tap][:, toks].numpy(), ref, rtol=2e-5, atol=2e-5)
        assert abs(taps[tap].abs().max().item() - g["absmax"][tap - 10]) <= 1e-3


def test_seeded_init_reproduces_reference_weights(golden):
    """Same RNG draws, same order as the reference's vit_base(patch_size=8) under seed 0:
    per-tensor sum / sum-of-squares / first / last element match the reference's."""
    g = golden("vit_init_seed0.npz")
    sd = _seed0_vit().state_dict()
    names = list(g["names"])
    assert sorted(sd.keys()) == names and len(names) == 150
    assert sum(v.numel() for v in sd.values()) == 85807872
    for name, chk, shape in zip(names, g["checks"], g["shapes"]):
        t = sd[name].double().flatten()
        assert str(tuple(sd[name].shape)) == shape
        got = np.array([t.sum().item(), (t * t).sum().item(), t[0].item(), t[-1].item()])
        np.testing.assert_array_equal(got, chk, err_msg=name)


def test_oracle_vit_matches_reference_forward(golden):
    from oracle import vit as ovit
    g = golden("vit_forward.npz")
    sd = _seed0_vit().state_dict()
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=11, scale=1.7))
    taps = ovit.forward_taps(sd, images, taps=(10, 11))
    toks = g["tokens"]
    for tap, key in ((10, "f10"), (11, "f11")):
        got = taps[tap][:, toks].numpy()
        np.testing.assert_allclose(got, g[key], rtol=1e-5, atol=1e-5)
        chk = g[key + "_check"]
        np.testing.assert_allclose(taps[tap].double().sum().item(), chk[0], rtol=0, atol=2e-2)
        np.testing.assert_allclose((taps[tap].double() ** 2).sum().item(), chk[1], rtol=1e-6)
    assert list(g["feature_dimensions"]) == [785, 785]          # token count, as upstream (common.py:274)
    assert abs(ovit.flops_per_image() / 1e9 - 156.3) < 0.5


# ---------------------------------------------------------------- aggregation
def test_oracle_aggregate_matches_reference_embed(golden):
    from oracle import aggregate as oagg, vit as ovit
    g = golden("aggregate_vit.npz")
    sd = _seed0_vit().state_dict()
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=11, scale=1.7))
    taps = ovit.forward_taps(sd, images, taps=(10, 11))
    Z, grids = oagg.aggregate([taps[10], taps[11]], 3, 2048, 4096)
    assert Z.shape == (2 * 784, 4096) and grids == [[28, 28], [28, 28]]
    np.testing.assert_allclose(Z[g["rows"]].numpy(), g["Z_rows"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(Z.double().sum(1).numpy(), g["Z_rowsum"], rtol=0, atol=5e-2)


@pytest.mark.parametrize("tag,salts,shape_a,shape_b,off,scales", [
    ("tok_small", (21, 22), (2, 37, 64), (2, 37, 64), (0.3, -0.1), (2.0, 1.0)),
    ("tok_p5", (21, 22), (2, 37, 64), (2, 37, 64), (0.3, -0.1), (2.0, 1.0)),
    ("cnn_small", (23, 24), (2, 32, 12, 12), (2, 48, 6, 6), (0.0, 0.0), (2.0, 1.5)),
])
def test_oracle_aggregate_variants(golden, tag, salts, shape_a, shape_b, off, scales):
    from oracle import aggregate as oagg
    g = golden("aggregate_%s.npz" % tag)
    fa = torch.from_numpy(gaussish_tensor(shape_a, salt=salts[0], scale=scales[0])) + off[0]
    fb = torch.from_numpy(gaussish_tensor(shape_b, salt=salts[1], scale=scales[1])) + off[1]
    Z, _ = oagg.aggregate([fa, fb], int(g["patchsize"]), int(g["pre"]), int(g["tgt"]))
    np.testing.assert_array_equal(Z.numpy(), g["Z"])      # same torch ops, same machine class: bit-exact


def test_oracle_aggregate_wrn_shapes(golden):
    from oracle import aggregate as oagg
    g = golden("aggregate_wrn.npz")
    wa = torch.from_numpy(gaussish_tensor((1, 512, 28, 28), salt=25, scale=2.0))
    wb = torch.from_numpy(gaussish_tensor((1, 1024, 14, 14), salt=26, scale=1.0))
    Z, grids = oagg.aggregate([wa, wb], 3, 1024, 1024)
    assert grids == [[28, 28], [14, 14]]
    np.testing.assert_array_equal(Z[g["rows"]].numpy(), g["Z_rows"])


def test_adaptive_windows():
    from oracle.aggregate import adaptive_windows
    s, e = adaptive_windows(6912, 2048)
    assert ((e - s) == 4).sum() == 1536 and ((e - s) == 5).sum() == 512
    s2, e2 = adaptive_windows(4096, 4096)
    assert np.array_equal(e2 - s2, np.ones(4096, dtype=np.int64))            # identity at 2*2048 -> 4096
    x = torch.from_numpy(gaussish_tensor((3, 6912), salt=5))
    ref = torch.nn.functional.adaptive_avg_pool1d(x.unsqueeze(1), 2048).squeeze(1).numpy()
    mine = np.stack([[x[r, a:b].double().mean().item() for a, b in zip(s, e)] for r in range(3)])
    np.testing.assert_allclose(mine, ref, rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------- alpha
def test_oracle_alpha_matches_reference(golden):
    from oracle import alpha as oa
    g = golden("alpha_small.npz")
    Zs, _ = clustered_embeddings(6, 49, 128, 3, salt=31)
    Zt, _ = clustered_embeddings(4, 49, 128, 1, salt=32)
    Z, Ztr = torch.from_numpy(Zs), torch.from_numpy(Zt)
    np.testing.assert_array_equal(oa.weights_unsupervised(Z).numpy(), g["w_unsup"])
    np.testing.assert_array_equal(oa.weights_supervised(Z, Ztr).numpy(), g["w_sup"])
    for tau in g["taus"]:
        np.testing.assert_array_equal(oa.matrix_alpha_unsupervised(float(tau), 1, Z).numpy(),
                                      g["alpha_unsup_%g" % tau])
        np.testing.assert_array_equal(oa.matrix_alpha_supervised(float(tau), 1, Z, Ztr).numpy(),
                                      g["alpha_sup_%g" % tau])
    a, X = oa.weighted_average(torch.from_numpy(g["alpha_unsup_1"]), Z)
    assert a.shape == (6, 1, 49) and a.dtype == torch.float32
    np.testing.assert_array_equal(X, g["X_unsup_1"])
    # tau = 0 with an exact tie: indicator / #ties
    Zs[0, 5] = Zs[0, 9]
    Zs[0, 5] += 40.0
    Zs[0, 9] = Zs[0, 5]
    np.testing.assert_array_equal(oa.matrix_alpha_unsupervised(0, 1, torch.from_numpy(Zs)).numpy(),
                                  g["alpha_tie_unsup_0"])
    assert abs(oa.FLOP_PER_PAIR(784, 4096) / 1e9 - 5.035) < 1e-3


def test_oracle_alpha_full_size(golden):
    from oracle import alpha as oa
    g = golden("alpha_full.npz")
    Zf, _ = clustered_embeddings(3, 784, 4096, 3, salt=41)
    Zb, _ = clustered_embeddings(2, 784, 4096, 1, salt=42)
    Z, Zbk = torch.from_numpy(Zf), torch.from_numpy(Zb)
    np.testing.assert_allclose(oa.weights_unsupervised(Z).numpy(), g["w_unsup"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(oa.weights_supervised(Z, Zbk).numpy(), g["w_sup"], rtol=0, atol=1e-4)


# ------------------------------------------------- cluster + metrics (KAT rows)
@pytest.mark.parametrize("name", ["kat_toothbrush_unsupervised", "kat_grid_unsupervised",
                                  "kat_toothbrush_supervised"])
def test_known_answer_rows_from_reference_csv(golden, name):
    """X and labels from the reference's committed pickles -> the NMI/ARI/F1 the reference's
    own CSV holds for that category at tau=2 (reference outputs/mvtec_ad/.../*_tau_result.csv)."""
    from oracle import cluster as oc
    g = golden(name + ".npz")
    NMI, ARI, F1, label, predict = oc.calculate_metrics(g["X"], list(g["anomaly"]))
    np.testing.assert_allclose([NMI, ARI, F1], g["expected"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(g["alpha_rowsum"], 1.0, rtol=0, atol=1e-5)
    assert len(label) == sum(a != "combined" for a in g["anomaly"])


def test_all_26_known_answer_rows_from_reference_csv(golden):
    """EVERY (alpha, X) artefact the reference commits (13 categories x 2 modes at tau = 2): the condensed distances
    of its kept images -> Ward -> cut -> Hungarian -> NMI / ARI / F1 = the reference's own CSV row, through the oracle.
    Where the full X is committed too (three rows), both routes must give the same labels."""
    from oracle import cluster as oc
    g = golden("kat_all_pdist.npz")
    rows = [str(r) for r in g["rows"]]
    assert len(rows) == 26
    for row in rows:
        cat, mode = row.rsplit("_", 1)
        names = [str(v) for v in g["names_" + cat]]
        d = g["pdist_" + row]
        assert d.dtype == np.float64 and d.shape == (len(names) * (len(names) - 1) // 2,) and "combined" not in names
        NMI, ARI, F1, label, predict, _ = oc.calculate_metrics_from_pdist(d, names)
        np.testing.assert_allclose([NMI, ARI, F1], g["expected_" + row], rtol=0, atol=1e-12, err_msg=row)
    for name in ("toothbrush_unsupervised", "grid_unsupervised", "toothbrush_supervised"):
        k = golden("kat_%s.npz" % name)
        _, _, _, label_x, predict_x = oc.calculate_metrics(k["X"], list(k["anomaly"]))
        cat = name.rsplit("_", 1)[0]
        _, _, _, label_d, predict_d, _ = oc.calculate_metrics_from_pdist(g["pdist_" + name], [str(v) for v in g["names_" + cat]])
        assert np.array_equal(label_x, label_d) and np.array_equal(predict_x, predict_d)


def test_best_map_is_a_permutation_fix():
    from oracle.cluster import best_map
    L1 = np.array([0, 0, 1, 1, 2, 2, 2])
    L2 = np.array([5, 5, 9, 9, 7, 7, 9])
    out = best_map(L1, L2)
    assert list(out) == [0, 0, 1, 1, 2, 2, 1]
