"""GPU: Ward linkage (nearest-neighbour chain) and the contingency-matrix scores on the device against what they
restate - scipy.cluster.hierarchy.linkage(y, "ward") and sklearn's NMI / ARI (reference
examples/test.py:193-220).

The linkage matrix must be BIT-IDENTICAL to scipy's on the same condensed distances: merge order, ids, heights
and sizes - on the reference's own committed X (the three known-answer fixtures), on N = 2 000 clustered points,
and on inputs full of exact ties (duplicated points, integer grids), where only the same tie rules (strict <,
previous chain element preferred, lowest index first, stable sort) give the same tree.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check_linkage(X, dev):
    import acnative
    from scipy.cluster import hierarchy
    from oracle.cluster import pdist
    n = X.shape[0]
    d = acnative.pdist(torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(dev))
    np.testing.assert_allclose(d.cpu().numpy(), pdist(X.astype(np.float32)), rtol=1e-12)
    Z = acnative.ward_linkage(d, n).cpu().numpy()
    ref = hierarchy.linkage(d.cpu().numpy(), method="ward")          # the SAME distances: bit for bit from here
    assert Z.shape == ref.shape == (n - 1, 4)
    assert np.array_equal(Z[:, :2], ref[:, :2]), "merge order / cluster ids differ"
    assert np.array_equal(Z[:, 3], ref[:, 3])
    assert np.array_equal(Z[:, 2], ref[:, 2]), "heights differ by %.3e" % np.abs(Z[:, 2] - ref[:, 2]).max()
    return Z


@pytest.mark.parametrize("name", ["kat_toothbrush_unsupervised", "kat_grid_unsupervised", "kat_toothbrush_supervised"])
def test_linkage_on_the_reference_embeddings(golden, dev, name):
    _check_linkage(golden(name + ".npz")["X"], dev)


def test_linkage_large_and_degenerate(dev):
    rng = np.random.default_rng(7)
    centers = rng.normal(size=(9, 48)) * 4.0
    X = centers[rng.integers(0, 9, 2000)] + rng.normal(size=(2000, 48))
    _check_linkage(X, dev)
    # exact ties: duplicated rows and an integer lattice
    Xd = np.repeat(rng.normal(size=(40, 8)), 3, axis=0)
    _check_linkage(Xd, dev)
    lattice = np.stack(np.meshgrid(np.arange(7), np.arange(6)), -1).reshape(-1, 2).astype(np.float64)
    _check_linkage(lattice, dev)
    for n in (2, 3, 5):
        _check_linkage(rng.normal(size=(n, 4)), dev)


def test_linkage_rejects_non_finite_distances(dev):
    """scipy raises ValueError on NaN / inf distances; the device chain must not walk off its arrays on them."""
    import acnative
    d = torch.ones(10, dtype=torch.float64, device=dev)               # n = 5
    for poison in (float("nan"), float("inf")):
        bad = d.clone()
        bad[3] = poison
        with pytest.raises(ValueError, match="finite"):
            acnative.ward_linkage(bad, 5)
    assert acnative.ward_linkage(d, 5).shape == (4, 4)
    assert acnative.ward_linkage(torch.empty(0, dtype=torch.float64, device=dev), 1).shape == (0, 4)


def test_scores_match_sklearn(dev):
    import acnative
    from sklearn import metrics
    rng = np.random.default_rng(11)
    cases = [(rng.integers(0, 6, 500), rng.integers(0, 6, 500)),                       # independent
             (np.arange(300) % 5, (np.arange(300) % 5 + (rng.random(300) < 0.2)) % 5),  # mostly right
             (np.arange(60) % 4, np.arange(60) % 4),                                     # perfect
             (np.zeros(30, int), np.zeros(30, int)),                                     # one class, one cluster
             (np.arange(40) % 3, np.zeros(40, int))]                                     # everything in one cluster
    for a, b in cases:
        ka, kb = int(a.max()) + 1, int(b.max()) + 1
        nmi, ari, C = acnative.cluster_scores(torch.from_numpy(a.astype(np.int32)).to(dev),
                                              torch.from_numpy(b.astype(np.int32)).to(dev), ka, kb)
        assert np.array_equal(C, metrics.cluster.contingency_matrix(a, b))
        assert abs(nmi - metrics.normalized_mutual_info_score(a, b)) <= 1e-12
        assert abs(ari - metrics.adjusted_rand_score(a, b)) <= 1e-12
    # many clusters (every image its own cluster and beyond the 48 KB static-LDS size: the sums live in dynamic LDS)
    a, b = rng.integers(0, 4000, 20000), rng.integers(0, 3500, 20000)
    nmi, ari, C = acnative.cluster_scores(torch.from_numpy(a.astype(np.int32)).to(dev), torch.from_numpy(b.astype(np.int32)).to(dev), 4000, 3500)
    assert abs(nmi - metrics.normalized_mutual_info_score(a, b)) <= 1e-10 and abs(ari - metrics.adjusted_rand_score(a, b)) <= 1e-10
    with pytest.raises(ValueError, match="outside"):
        acnative.cluster_scores(torch.tensor([0, 7], dtype=torch.int32, device=dev),
                                torch.tensor([0, 1], dtype=torch.int32, device=dev), 2, 2)


@pytest.mark.parametrize("name", ["kat_toothbrush_unsupervised", "kat_grid_unsupervised", "kat_toothbrush_supervised"])
def test_known_answer_rows_end_to_end_on_the_device(golden, dev, name):
    """pdist -> Ward -> cut -> relabel -> NMI / ARI / F1 on the reference's committed X reproduces the reference's
    CSV rows (the CPU twin of this test replaces the three device entry points by scipy / sklearn)."""
    from patchcore.clustering import cluster_metrics
    g = golden(name + ".npz")
    NMI, ARI, F1, label, predict = cluster_metrics(g["X"], list(g["anomaly"]), device=dev)
    np.testing.assert_allclose([NMI, ARI, F1], g["expected"], rtol=0, atol=1e-12)


def test_all_26_known_answer_rows_on_the_device(golden, dev):
    """every (alpha, X) artefact the reference commits - 13 categories x {unsupervised, supervised} - from the
    condensed float64 distances of its kept images (tests/golden/kat_all_pdist.npz) through ac_ward_linkage -> cut -> relabel
    -> ac_cluster_scores: the reference's CSV row to 1e-12, and the linkage matrix BIT-IDENTICAL to scipy's on the same
    distances (merge order, ids, heights, sizes) - the Ward kernel's tie rules on 26 sets of real embeddings."""
    import acnative
    from scipy.cluster import hierarchy
    from patchcore.clustering import cluster_metrics_from_pdist
    g = golden("kat_all_pdist.npz")
    rows = [str(r) for r in g["rows"]]
    assert len(rows) == 26
    for row in rows:
        cat, mode = row.rsplit("_", 1)
        names = [str(v) for v in g["names_" + cat]]
        d_host = g["pdist_" + row]
        d = torch.from_numpy(d_host).to(dev)
        Z = acnative.ward_linkage(d, len(names)).cpu().numpy()
        assert np.array_equal(Z, hierarchy.linkage(d_host, method="ward")), row
        NMI, ARI, F1, label, predict = cluster_metrics_from_pdist(d, names)
        np.testing.assert_allclose([NMI, ARI, F1], g["expected_" + row], rtol=0, atol=1e-12, err_msg=row)
