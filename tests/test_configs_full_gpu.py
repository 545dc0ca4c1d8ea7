"""GPU: every BASELINE.json config at its STATED size (VERDICT r02 "Next round" item 2).

  2   bottle-shaped category, N = 83, unsupervised tau = 1: the WHOLE alpha matrix, X and NMI / ARI / F1 against the
      CPU oracle run on the same images and weights (about a minute of host time: 83 ViT forwards at batch 1 and
      6 806 ordered cdist pairs - utils.py:240-257, test.py:176-220).  Gates: alpha rel 1e-4 / abs 1e-6, X 1e-4,
      metrics +-1e-3 (north_star's bar).
  3   all 15 MVTec-AD test-set sizes (83 ... 167, sum 1725) through run_category: alpha rows sum to 1, w of two images
      per category against the float64 brute force, sub-blocked schedule == single launch to the bit.
  4   supervised, bank M = N_test, on four category sizes including the largest (N = M = 167): w against the float64
      brute force over the whole bank, pruned search == exhaustive blocks.
  5b  one synthetic category of N = 2000 (2.0e16 FLOP, 10 block launches): w of three images against the float64
      brute force over all 1999 key images, and the 64-bit fixed-point scale of the sum checked against its bound.
  5a  N = 10 000 on one GPU (`average` weights): X of sampled images against the CPU oracle, the 10k x 10k condensed
      distances against float64 numpy on a 1000-image subsample, and the WHOLE 10 000-leaf Ward linkage bit-identical
      to scipy's on the same distances.

The float64 brute force is test-side arithmetic (torch fp64 GEMMs on the GPU: |a|^2 + |b|^2 - 2ab in fp64 is 1e-12
from the truth at these magnitudes); the oracle legs run on the host CPU.
"""
import math
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W_TOL = 2e-5          # |w - fp64 truth|: the distance of the reference's own fp32 cdist from the truth
P, D = 784, 4096


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


@pytest.fixture(autouse=True)
def _free_device_memory():
    """The large cases need most of the HBM: give back what the previous test cached."""
    import acnative
    yield
    acnative.release_workspaces()
    torch.cuda.empty_cache()


def _truth_w(Zq_rows, Zk, reduce, skip=None, chunk=48):
    """float64 w of the query images Zq_rows [m, P, D] against every key image of Zk [n, P, D] (fp32 on the device),
    key images in chunks (an fp64 copy of a 2000-image Z would be 51 GB).  reduce 0: mean over the key images other
    than skip[i] (utils.py:222-227); reduce 1: min over all of them (utils.py:230-237)."""
    m = Zq_rows.shape[0]
    q = Zq_rows.double()
    qsq = (q * q).sum(2)                                            # [m, P]
    tot = torch.zeros((m, Zq_rows.shape[1]), dtype=torch.float64, device=Zk.device)
    best = torch.full_like(tot, float("inf"))
    cnt = [0] * m
    for j0 in range(0, Zk.shape[0], chunk):
        k = Zk[j0:j0 + chunk].double()
        ksq = (k * k).sum(2)                                        # [c, P]
        for i in range(m):
            d2 = qsq[i][None, :, None] + ksq[:, None, :] - 2.0 * torch.matmul(q[i][None], k.transpose(1, 2))
            dmin = d2.clamp_min(0).sqrt().min(2)[0]                 # [c, P]: min over the key image's patches
            if reduce == 0:
                for jj in range(dmin.shape[0]):
                    if skip is not None and j0 + jj == skip[i]:
                        continue
                    tot[i] += dmin[jj]
                    cnt[i] += 1
            else:
                best[i] = torch.minimum(best[i], dmin.min(0)[0])
    if reduce == 0:
        return tot / torch.tensor(cnt, dtype=torch.float64, device=Zk.device)[:, None]
    return best


# ------------------------------------------------------------------------------------------------ config 2
def test_config2_whole_alpha_matrix_against_oracle(dev, core, config2_oracle):
    """BASELINE config 2 (the bench workload): every one of the 83 x 784 alpha values, X and the three scores.  The oracle
    leg (a minute of host time) was started as a CPU-only child process when the session began (conftest.py,
    tests/_config2_oracle.py) and is collected here; it counts only if the child's weights ARE the device's weights."""
    import _config2_oracle as oleg
    from oracle import cluster as ocluster
    from patchcore.pipeline import HipOps, run_category
    from patchcore.datasets import synthetic
    from patchcore.clustering import cluster_metrics_from_pdist
    labels = synthetic.category_labels("bottle")
    images = synthetic.make_images("bottle", labels)
    n = len(labels)
    assert n == 83

    out = run_category(HipOps(core, batch_size=128), images.to(dev), "unsupervised", tau=1.0)
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    sd = {k: v.detach().cpu() for k, v in core.backbone.state_dict().items()}
    res = config2_oracle()
    if res is not None and not np.array_equal(res["weight_sums"], oleg.weight_sums(sd).numpy()):
        res = None                                               # (cannot happen with the seeded init: then nothing of it is used)
    if res is None:                                              # no child (a single-test run on a box without one): here and now
        threads = torch.get_num_threads()
        torch.set_num_threads(min(32, os.cpu_count() or 1))      # the box's default (one per hardware thread) is 4x slower here
        res = oleg.compute(sd, images, labels)
        torch.set_num_threads(threads)
        print("oracle at N=83 computed inside the test: %.1f s of host time" % (time.perf_counter() - t0))
    else:
        print("oracle at N=83: %.1f s of host time in the child process, %.1f s waited for here"
              % (float(res["seconds"][0]), time.perf_counter() - t0))
    Zr, w_ref = torch.from_numpy(res["Z"]), torch.from_numpy(res["w"])
    a_ref, X_ref = torch.from_numpy(res["alpha"]), res["X"]
    rn, ra, rf = (float(v) for v in res["scores"])
    rlabel = res["label"]

    a = out["alpha"].cpu().numpy()
    assert a.shape == (n, 1, P)
    np.testing.assert_allclose(a, a_ref.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["X"], X_ref, rtol=0, atol=1e-4)
    Zg = out["Z_loc"].cpu()
    assert (Zg - Zr).abs().max().item() <= 3e-6 + 2e-6 * Zr.abs().max().item()
    # w itself: the oracle's fp32 cdist and the split-fp16 kernel are each within W_TOL of the float64 truth
    assert (out["w_loc"].cpu() - w_ref).abs().max().item() <= 2 * W_TOL
    nmi, ari, f1, label, predict = cluster_metrics_from_pdist(out["pdist"], labels)
    print("N=83: alpha max rel err %.2e, X max abs err %.2e, NMI/ARI/F1 %.4f/%.4f/%.4f (oracle %.4f/%.4f/%.4f)"
          % (np.max(np.abs(a - a_ref.numpy()) / np.maximum(a_ref.numpy(), 1e-12)), np.max(np.abs(out["X"] - X_ref)),
             nmi, ari, f1, rn, ra, rf))
    assert abs(nmi - rn) <= 1e-3 and abs(ari - ra) <= 1e-3 and abs(f1 - rf) <= 1e-3
    assert np.array_equal(label, rlabel)


# ------------------------------------------------------------------------------------------------ config 3
def test_config3_all_fifteen_category_sizes(dev, core):
    """One pass over all 15 categories at the real test-set sizes, unsupervised tau = 1."""
    from patchcore.pipeline import HipOps, run_category
    from patchcore.datasets import synthetic
    ops = HipOps(core, batch_size=128)
    sizes, worst = [], 0.0
    for cat in synthetic.TEST_HISTOGRAMS:
        labels = synthetic.category_labels(cat)
        n = len(labels)
        sizes.append(n)
        images = synthetic.make_images(cat, labels).to(dev)
        out = run_category(ops, images, "unsupervised", tau=1.0, want_host=False)
        alpha = out["alpha"].squeeze(1)
        assert alpha.shape == (n, P) and torch.isfinite(alpha).all() and (alpha > 0).all()
        assert (alpha.double().sum(1) - 1.0).abs().max().item() <= 1e-5
        assert out["pdist"].shape == (n * (n - 1) // 2,) and torch.isfinite(out["pdist"]).all()
        Z = out["Z_loc"]
        idx = [1, n - 2]
        truth = _truth_w(Z[idx], Z, 0, skip=idx)
        err = (out["w_loc"][idx].double() - truth).abs().max().item()
        worst = max(worst, err)
        assert err <= W_TOL, "%s N=%d: |w - fp64 truth| = %.2e" % (cat, n, err)
        # the sub-blocked ring schedule on the same embeddings: 3-4 blocks per category, same bits
        blocked = run_category(ops, images, "unsupervised", tau=1.0, want_host=False, want_pdist=False,
                               budget=48 * 48 * P * 4, Z_loc=Z)
        assert torch.equal(blocked["w_loc"], out["w_loc"]), "%s: blocked schedule differs from the single launch" % cat
        del out, blocked, Z, images
    assert sizes == [83, 150, 132, 117, 78, 110, 124, 115, 167, 160, 117, 42, 100, 79, 151] and sum(sizes) == 1725
    print("config 3: worst |w - fp64 truth| over 15 categories = %.2e" % worst)


# ------------------------------------------------------------------------------------------------ config 4
@pytest.mark.parametrize("category", ["toothbrush", "bottle", "cable", "pill"])
def test_config4_supervised_category_sizes(dev, core, category):
    """Supervised alpha with a bank of M = N_test seeded train images (main.py:274-282), N = M = 42 / 83 / 150 / 167:
    the pruned exact search against the float64 brute force over the whole bank AND against the exhaustive blocks."""
    from patchcore.pipeline import HipOps, run_category, select_bank
    from patchcore.datasets import synthetic
    labels = synthetic.category_labels(category)
    n = len(labels)
    ntr = synthetic.TRAIN_SIZES[category]
    images = synthetic.make_images(category, labels).to(dev)
    train = synthetic.make_images(category, ["good"] * ntr, split=synthetic.DatasetSplit.TRAIN).to(dev)
    ops = HipOps(core, batch_size=128)
    out = run_category(ops, images, "supervised", tau=1.0, train_images=train, train_ratio=1.0, want_host=False)
    alpha = out["alpha"].squeeze(1)
    assert alpha.shape == (n, P) and torch.isfinite(alpha).all()
    assert (alpha.double().sum(1) - 1.0).abs().max().item() <= 1e-5
    sel = select_bank(ntr, n, 1.0, 0)
    assert len(sel) == min(ntr, n)
    bank = ops.embed(train[sel.to(dev)])
    idx = [0, n // 2, n - 1]
    truth = _truth_w(out["Z_loc"][idx], bank, 1)
    err = (out["w_loc"][idx].double() - truth).abs().max().item()
    print("supervised %s N=%d M=%d: |w - fp64 truth| = %.2e" % (category, n, len(sel), err))
    assert err <= W_TOL
    exhaustive = run_category(ops, images, "supervised", tau=1.0, train_images=train, train_ratio=1.0, want_host=False,
                              want_pdist=False, prune=False, Z_loc=out["Z_loc"])
    # the pruned search re-measures the nearest candidate in fp64, the exhaustive blocks in split fp16: both within
    # W_TOL of the truth, and of each other
    assert (exhaustive["w_loc"] - out["w_loc"]).abs().max().item() <= W_TOL


# ------------------------------------------------------------------------------------------------ config 5b
def test_config5b_two_thousand_images(dev, core):
    """N = 2000 synthetic images, unsupervised: the sub-blocked schedule at a size where it is the only way to run
    (4 symmetric + 6 two-way launches of <= 585 images), w against the float64 brute force, fixed-point scale."""
    import acnative
    from patchcore.pipeline import HipOps, run_category, block_images
    from patchcore.datasets import synthetic
    n = 2000
    kinds = ["good", "a", "b", "c", "d"]
    labels = [kinds[(i * len(kinds)) // n] for i in range(n)]
    images = synthetic.make_images("synthetic", labels)          # stays on the host: embed_device moves it batch by batch
    ops = HipOps(core, batch_size=128)
    assert block_images(P) == 585
    acnative.kernel_events = []
    try:
        out = run_category(ops, images, "unsupervised", tau=1.0, want_host=False)
        torch.cuda.synchronize()
        events = acnative.kernel_events
    finally:
        acnative.kernel_events = None
    assert len(events) == 10                                      # 4 symmetric + 6 two-way block launches
    assert abs(sum(e[3] for e in events) - n * (n - 1.0)) < 1e-3   # every ordered image pair exactly once
    alpha = out["alpha"].squeeze(1)
    assert alpha.shape == (n, P) and torch.isfinite(alpha).all() and (alpha > 0).all()
    assert (alpha.double().sum(1) - 1.0).abs().max().item() <= 1e-5
    Z = out["Z_loc"]
    idx = [0, 777, n - 1]                                         # first block, a middle block, last (ragged) block
    truth = _truth_w(Z[idx], Z, 0, skip=idx)
    err = (out["w_loc"][idx].double() - truth).abs().max().item()
    print("N=2000: |w - fp64 truth| = %.2e (w ~ %.1f)" % (err, truth.mean().item()))
    assert err <= W_TOL
    # the 64-bit fixed point of the sum: shift = min(40, 62 - ceil(log2(N * 2 sqrt(max |z|^2) + 1))) (csrc/alpha.hip);
    # the largest possible sum, N * 2 sqrt(max |z|^2) * 2^shift, stays below 2^62, and one unit is far below fp32's ulp
    m = torch.zeros(1, dtype=torch.float32, device=dev)
    for s in range(0, n, 250):
        m = torch.maximum(m, acnative.max_sq([ops.pack(Z[s:s + 250])], dev))
    shift = int(acnative.fixed_shift(m, n).item())
    bound = n * 2.0 * math.sqrt(float(m.item())) + 1.0
    assert shift == min(40, 62 - math.frexp(bound)[1])
    assert bound * 2.0 ** shift < 2.0 ** 62 and 2.0 ** -shift < 1e-9


def test_config5b_full_size_schedule_with_the_launches_as_bookkeeping(dev, core):
    """Config 5b at its FULL size, N = 10 000 on one GPU, under the driver's eyes without the 544 s step (VERDICT r05 item 6;
    the checked full run: scratch/verify_5b_n10000.py, profiles/r05_verify_5b_n10000.json).  Everything that DECIDES how
    that run goes is the product's own code on this box - the lean-mode decision from the free HBM, the block list
    (pipeline.unsup_schedule through alpha_weights_unsupervised), the budget of a matrix of minima, the int64 accumulator
    and the fixed-point scale from the library - only the MFMA launches and their folds are replaced by bookkeeping:
      * lean mode is chosen (Z + packed of 10 000 images: 257 GB) and what it keeps instead (taps + packed + two
        matrices of minima + sums: 177 GB) fits the free HBM; N = 83 is not lean,
      * 171 block launches (18 symmetric + 153 two-way, <= 585 images a side), each matrix of minima inside the budget,
      * every ORDERED image pair is folded into the sums exactly once, no image against itself,
      * the scale: 9 999 sums of at most 2 sqrt(max |z|^2) * 2^shift stay below 2^62."""
    import acnative
    from patchcore.pipeline import HipOps, alpha_weights_unsupervised, shard_bounds, block_images, MINBUF_BYTES, MEAN, NO_SELF
    n, D = 10000, 4096
    max_sq_value = 2600.0                                          # |z|^2 of a DINO-shaped embedding row (config 2: ~2.4e3)

    class Block:                                                   # what a packed block is to the schedule: a range of images
        def __init__(self, lo, hi):
            self.lo, self.hi, self.n, self.P, self.D = lo, hi, hi - lo, P, D

        def slice(self, a, b):
            assert 0 <= a <= b <= self.n
            return Block(self.lo + a, self.lo + b)

    class Minima:
        def __init__(self, keys, queries):
            self.keys, self.queries, self.shape = keys, queries, (keys.n, queries.n * P)

    cov = np.zeros((n, n), dtype=np.uint8)                         # [key image, query image]: folds into the query's sum
    launches = []

    class BookOps(HipOps):
        def max_sq(self, packed_list, device, out=None):
            return torch.full((1,), max_sq_value, dtype=torch.float32, device=device)

        def pair_block(self, keys, queries, two_way=False, part=0, nparts=1):
            sym = keys.lo == queries.lo and keys.hi == queries.hi
            assert keys.n * queries.n * P * 4 <= MINBUF_BYTES and max(keys.n, queries.n) <= 585
            assert sym != two_way and (part, nparts) == (0, 1)
            launches.append((keys.n * (keys.n - 1.0)) if sym else 2.0 * keys.n * queries.n)
            return (Minima(keys, queries), Minima(queries, keys)) if two_way else Minima(keys, queries)

        def pair_accumulate(self, mb, P_, reduce, self_first, shift, acc):
            k, q = mb.keys, mb.queries
            assert P_ == P and reduce == MEAN and acc.numel() == q.n * P and acc.dtype == torch.int64
            assert acc.data_ptr() == acc_base[0] + q.lo * P * 8        # the query images' own slots of the one accumulator
            assert int(shift.item()) == expect_shift
            cov[k.lo:k.hi, q.lo:q.hi] += 1
            if self_first != NO_SELF:                              # symmetric block: column c's own image is key self_first + c // P
                assert self_first == 0 and (k.lo, k.hi) == (q.lo, q.hi)
                i = np.arange(k.lo, k.hi)
                cov[i, i] -= 1

        def new_acc(self, ncols, reduce, device):
            t = HipOps.new_acc(self, ncols, reduce, device)
            acc_base[0] = t.data_ptr()
            return t

    ops = BookOps(core, batch_size=256)
    # ---- the lean decision and the memory plan of the lean run, on THIS box's free HBM
    acnative.release_workspaces()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(dev)
    assert ops.wants_lean(n) and not ops.wants_lean(83)
    # ... and the shortcut by which a multi-rank job skips the agreement on it altogether: a function of the shapes and the device
    # model only (216 images: a rank's share of all 15 categories on 8 GPUs; 1 400 images are an eighth of the HBM)
    assert ops.never_lean(216) and ops.never_lean(1300) and not ops.never_lean(1725) and not ops.never_lean(n)
    tap_bytes = sum(int(np.prod(shape)) * torch.empty((), dtype=dt).element_size() for shape, dt in ops._tapshapes)
    assert tap_bytes == 2 * 785 * 768 * 4
    lean_bytes = n * tap_bytes + n * P * (D * 4 + 4) + 2 * MINBUF_BYTES + n * P * 8
    full_bytes = n * 2 * P * D * 4
    print("N=10000: Z + packed %.0f GB, lean %.0f GB, free %.0f of %.0f GB" % (full_bytes / 1e9, lean_bytes / 1e9, free / 1e9, total / 1e9))
    assert lean_bytes < 0.9 * free and full_bytes > HipOps.LEAN_FRACTION * free
    # ---- the schedule
    assert block_images(P) == 585
    bound = n * 2.0 * math.sqrt(max_sq_value) + 1.0
    expect_shift = min(40, 62 - math.frexp(bound)[1])
    acc_base = [0]
    w = alpha_weights_unsupervised(ops, Block(0, n), shard_bounds(n, 1), 0, 1, None, dev)
    torch.cuda.synchronize()
    assert w.shape == (n, P) and w.dtype == torch.float32
    assert len(launches) == 171 and launches.count(585 * 584.0) == 17          # 18 symmetric (the last one ragged) + 153 two-way
    assert abs(sum(launches) - n * (n - 1.0)) < 1e-3
    off = ~np.eye(n, dtype=bool)
    assert (cov[off] == 1).all() and (np.diagonal(cov) == 0).all()
    assert bound * 2.0 ** expect_shift < 2.0 ** 62 and 2.0 ** -expect_shift < 1e-9 and expect_shift == int(acnative.fixed_shift(
        torch.full((1,), max_sq_value, dtype=torch.float32, device=dev), n).item())


# ------------------------------------------------------------------------------------------------ config 5a
def test_config5a_ten_thousand_images(dev, core):
    """N = 10 000 on ONE GPU, `average` weights (extract + X + 10k x 10k distances + Ward): Z alone is 128 GB of the
    288 GB.  X of sampled images against the CPU oracle; the distances against float64 numpy on 1000 of the images;
    the 9 999-row linkage matrix bit-identical to scipy's from the same distances."""
    import acnative
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import pdist as scipy_pdist
    from oracle import vit as ovit, aggregate as oagg
    from patchcore.pipeline import HipOps, run_category
    from patchcore.datasets import synthetic
    from patchcore.clustering import cluster_metrics_from_pdist
    n = 10000
    kinds = ["good", "a", "b", "c", "d"]
    labels = [kinds[(i * len(kinds)) // n] for i in range(n)]
    images = synthetic.make_images("synthetic", labels)          # 6 GB on the host, moved batch by batch
    ops = HipOps(core, batch_size=128)
    t0 = time.perf_counter()
    out = run_category(ops, images, "average", want_host=True)
    torch.cuda.synchronize()
    print("N=10000 average: %.1f s on the GPU" % (time.perf_counter() - t0))
    X = out["X"]
    assert X.shape == (n, D) and np.isfinite(X).all()
    alpha = out["alpha"]
    assert alpha.shape == (n, 1, P) and torch.all(alpha == alpha[0, 0, 0]) and abs(alpha[0, 0, 0].item() * P - 1) < 1e-6
    out.pop("Z_loc")
    acnative.release_workspaces()
    torch.cuda.empty_cache()

    sd = {k: v.detach().cpu() for k, v in core.backbone.state_dict().items()}
    for i in (0, 4999, n - 1):                                    # main.py:290-296 with alpha = 1/P: X_i = mean_p Z_i[p]
        taps = ovit.forward_taps(sd, images[i:i + 1], taps=(10, 11))
        Zi = oagg.aggregate([taps[10], taps[11]], 3, 2048, 4096)[0].reshape(P, D)
        xi = torch.bmm(torch.full((1, 1, P), 1.0 / P), Zi[None]).reshape(D).numpy()
        np.testing.assert_allclose(X[i], xi, rtol=0, atol=1e-5)

    cond = out["pdist"]
    assert cond.shape == (n * (n - 1) // 2,)
    sub = np.arange(0, n, 10)                                     # 1000 images spread over the five classes
    iu = np.triu_indices(len(sub), 1)
    a, b = sub[iu[0]].astype(np.int64), sub[iu[1]].astype(np.int64)
    pos = torch.from_numpy(a * n - a * (a + 1) // 2 + (b - a - 1)).to(dev)
    got = cond[pos].cpu().numpy()
    np.testing.assert_allclose(got, scipy_pdist(X[sub].astype(np.float64)), rtol=1e-10, atol=0)

    t0 = time.perf_counter()
    Zl = acnative.ward_linkage(cond, n).cpu().numpy()
    t1 = time.perf_counter()
    Zs = linkage(cond.cpu().numpy(), "ward")                      # what sklearn's ward_tree calls (test.py:193-195)
    print("Ward N=10000: device %.2f s, scipy %.2f s" % (t1 - t0, time.perf_counter() - t1))
    assert np.array_equal(Zl, Zs), "the 10 000-leaf linkage matrix differs from scipy's"
    nmi, ari, f1, _, _ = cluster_metrics_from_pdist(cond, labels)
    assert 0.0 <= nmi <= 1.0 and -1.0 <= ari <= 1.0 and 0.0 <= f1 <= 1.0
