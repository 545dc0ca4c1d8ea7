"""CPU: host-side logic of the drop-in (tree cut, label matching, sharding, path / CSV / pickle
formats, datasets).  Where a function needs the GPU pdist, the test swaps `acnative.pdist` for the
oracle's (tests are the only place that is allowed)."""
import csv
import os
import sys

import numpy as np
import pytest
import torch

from conftest import EXAMPLES


@pytest.fixture()
def oracle_pdist(monkeypatch):
    """No GPU here: the three device entry points of the clustering stage are replaced by what they restate
    (scipy pdist, scipy ward linkage, sklearn NMI / ARI) so that the HOST logic around them can be tested.
    tests/test_cluster_gpu.py checks the device versions against exactly these."""
    import acnative
    from oracle import cluster as oc
    from scipy.cluster import hierarchy
    from sklearn import metrics
    monkeypatch.setattr(acnative, "pdist", lambda X: torch.from_numpy(oc.pdist(X.cpu().numpy())))
    monkeypatch.setattr(acnative, "ward_linkage",
                        lambda c, n: torch.from_numpy(hierarchy.linkage(c.cpu().numpy(), method="ward")))
    monkeypatch.setattr(acnative, "cluster_scores", lambda a, b, ka, kb: (
        metrics.normalized_mutual_info_score(a.numpy(), b.numpy()), metrics.adjusted_rand_score(a.numpy(), b.numpy()), None))


def _examples():
    if EXAMPLES not in sys.path:
        sys.path.insert(0, EXAMPLES)
    sys.modules.pop("test", None)
    import importlib
    main = importlib.import_module("main")
    spec = importlib.util.spec_from_file_location("ac_examples_test", os.path.join(EXAMPLES, "test.py"))
    test_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(test_mod)
    return main, test_mod


def test_cut_tree_equals_sklearn_cut(oracle_pdist):
    from sklearn.cluster._agglomerative import _hc_cut
    from scipy.cluster import hierarchy
    from patchcore.clustering import cut_tree, ward_children
    from oracle.cluster import pdist
    rng = np.random.default_rng(3)
    for n, k in [(2, 1), (2, 2), (7, 3), (40, 6), (41, 41)]:
        X = rng.normal(size=(n, 5))
        children = ward_children(torch.from_numpy(pdist(X)), n)
        ref_children = hierarchy.ward(X)[:, :2].astype(np.intp)
        assert np.array_equal(children, ref_children)
        assert np.array_equal(cut_tree(children, k, n), _hc_cut(k, ref_children, n))
    with pytest.raises(ValueError):
        cut_tree(children, n + 1, n)


@pytest.mark.parametrize("name", ["kat_toothbrush_unsupervised", "kat_grid_unsupervised", "kat_toothbrush_supervised"])
def test_cluster_metrics_known_answer_rows(golden, oracle_pdist, name):
    """Product host logic (own cut / relabel / scipy linkage on a condensed matrix) on the reference's
    committed X reproduces the reference's CSV rows."""
    from patchcore.clustering import cluster_metrics
    g = golden(name + ".npz")
    NMI, ARI, F1, label, predict = cluster_metrics(g["X"], list(g["anomaly"]), device=torch.device("cpu"))
    np.testing.assert_allclose([NMI, ARI, F1], g["expected"], rtol=0, atol=1e-12)


def test_best_map_matches_oracle():
    from patchcore.clustering import best_map
    from oracle.cluster import best_map as ref
    rng = np.random.default_rng(5)
    for _ in range(20):
        k = int(rng.integers(2, 7))
        L1, L2 = rng.integers(0, k, 60), rng.integers(0, k, 60) + 10
        assert np.array_equal(best_map(L1, L2), ref(L1, L2))


def test_label_encoding_and_micro_f1_equal_sklearn():
    """clustering.encode_labels / micro_f1 replace sklearn's LabelEncoder / f1_score(average="micro") (what the reference
    calls, examples/test.py:190-191, 220) on the scoring path: same ids, same score - also when a predicted id never
    occurs in the truth, for one class, and for strings that sort unlike they were met."""
    from sklearn import metrics
    from sklearn.preprocessing import LabelEncoder
    from patchcore.clustering import encode_labels, micro_f1
    rng = np.random.default_rng(3)
    for names in (["good", "crack", "good", "broken_large", "crack"], ["b"] * 4, ["z", "a", "m", "a"], []):
        ref = LabelEncoder().fit_transform(names) if names else np.zeros(0, dtype=int)
        assert np.array_equal(encode_labels(names), ref)
    for k_true, k_pred, n in ((4, 4, 83), (6, 9, 160), (2, 3, 42), (1, 1, 5), (3, 3, 1)):
        for _ in range(20):
            label, predict = rng.integers(0, k_true, n), rng.integers(0, k_pred, n)
            assert abs(micro_f1(label, predict) - metrics.f1_score(label, predict, average="micro")) <= 1e-15
    assert micro_f1(np.array([0, 1, 2]), np.array([1, 2, 0])) == 0.0


def test_shard_bounds_and_bank():
    from patchcore.pipeline import shard_bounds, select_bank
    assert shard_bounds(83, 8) == [(0, 11), (11, 22), (22, 33), (33, 44), (44, 55), (55, 66), (66, 77), (77, 83)]
    assert shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert shard_bounds(10, 1) == [(0, 10)]
    for n, w in [(1725, 8), (167, 8), (42, 5), (0, 2)]:
        b = shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    a, b = select_bank(209, 83, 1.0, seed=0), select_bank(209, 83, 1.0, seed=0)
    assert torch.equal(a, b) and len(a) == 83 and len(set(a.tolist())) == 83       # N_test, not N_train (main.py:281)
    assert len(select_bank(60, 42, 2.0)) == 60 and len(select_bank(60, 42, 0.5)) == 21


def test_unsupervised_block_schedule_covers_every_pair_once():
    """Ring schedule with symmetric halving (pipeline.unsup_schedule): over all ranks, every ordered pair of
    distinct images - and every image against itself inside its symmetric block - is produced exactly once,
    for shard counts that do and do not divide N, empty shards, even world sizes (the split step G/2) and
    sub-block sizes from 1 image to 'everything'."""
    from patchcore.pipeline import unsup_schedule, shard_bounds, block_images
    for n in (0, 1, 2, 3, 7, 16, 83):
        for world in (1, 2, 3, 4, 5, 8):
            for c in (1, 2, 5, 1000):
                b = shard_bounds(n, world)
                cov = np.zeros((n, n), int)
                for r in range(world):
                    for it in unsup_schedule(b, r, c):
                        (a0, a1) = it[2]
                        assert b[r][0] <= a0 and a1 <= b[r][1] and a1 - a0 <= c       # keys are always local
                        if it[0] == "sym":
                            cov[a0:a1, a0:a1] += 1
                        else:
                            (b0, b1) = it[3]
                            src = (r + it[1]) % world
                            assert b[src][0] <= b0 and b1 <= b[src][1] and b1 - b0 <= c
                            cov[a0:a1, b0:b1] += 1
                            cov[b0:b1, a0:a1] += 1
                assert (cov == 1).all(), (n, world, c)
    assert block_images(784, 1 << 30) == 585 and block_images(784, 100) == 1


def test_later_ring_steps_fuse_into_rectangles():
    """pipeline.merge_later_steps: the blocks of ring steps >= 1 fused into as few two-way rectangles as their shapes
    allow.  Over all ranks the fused launches still cover every pair of images from different shards exactly once; a fused
    block never crosses image 0 (its remote images are one ascending range on ONE side of the local block - the
    orientation rule), stays inside the block limits, and the count of launches is what DESIGN.md section 6 says: ONE per rank of
    the lower half of an even world, at most three in the upper half."""
    from patchcore.pipeline import unsup_schedule, merge_later_steps, shard_bounds
    P = 784
    for n in (5, 16, 83, 167):
        for world in (2, 3, 4, 5, 8):
            for c in (3, 1000):
                b = shard_bounds(n, world)
                cov = np.zeros((n, n), int)
                for r in range(world):
                    items = [it for it in unsup_schedule(b, r, c) if it[1] >= 1]
                    fused = merge_later_steps(items, c, P)
                    for _, (k1, k2), (a0, a1), (b0, b1) in fused:
                        assert 1 <= k1 <= k2 <= world // 2 and b[r][0] <= a0 < a1 <= b[r][1]
                        assert b1 - b0 <= c and a1 - a0 <= c
                        # the remote images: one ascending range starting inside the shard of step k1, ending inside that of k2
                        s1, s2 = b[(r + k1) % world], b[(r + k2) % world]
                        assert s1[0] <= b0 < s1[1] and s2[0] < b1 <= s2[1] and b0 < b1
                        assert (b0 >= a1) or (b1 <= a0)                      # wholly on one side of the local block
                        cov[a0:a1, b0:b1] += 1
                        cov[b0:b1, a0:a1] += 1
                    if c == 1000 and n >= world:
                        assert len(fused) <= (1 if (world % 2 == 0 and r < world // 2) or world <= 3 else 3), (n, world, r, fused)
                for r in range(world):                                       # what step 0 covers: pairs inside one shard
                    cov[b[r][0]:b[r][1], b[r][0]:b[r][1]] += 1
                assert (cov == 1).all(), (n, world, c)
    # a fused block that would not fit the matrix-of-minima budget is left in pieces
    items = [("two", 1, (0, 10), (10, 20)), ("two", 2, (0, 10), (20, 30))]
    assert len(merge_later_steps(items, 1000, P)) == 1 and len(merge_later_steps(items, 1000, P, budget=10 * 15 * P * 4)) == 2


def test_artefact_paths_match_reference_scheme():
    main, test_mod = _examples()
    d = main.artefact_dir("outputs/mvtec_ad/dino_vitbase8/unsupervised", ["blocks.10", "blocks.11"], 2048, 4096, 2, 1)
    assert d == "outputs/mvtec_ad/dino_vitbase8/unsupervised/blocks.10_blocks.11_2048_4096_2.0_1.0"
    p = test_mod.artefact_path("bottle", 2048, 4096, ["dino_vitbase8"], ["blocks.10", "blocks.11"], 1, 2,
                               "unsupervised", "mvtec_ad")
    assert p == ("../outputs/mvtec_ad/dino_vitbase8/unsupervised/blocks.10_blocks.11_2048_4096_2.0_1.0/"
                 "matrix_alpha_X_bottle_unsupervised.pickle")
    args = main.build_parser().parse_args([])
    assert (args.path, args.backbone_names, args.layers_to_extract_from, args.pretrain_embed_dimension,
            args.target_embed_dimension, args.output_dir, args.patchsize, args.tau, args.train_ratio,
            args.supervised, args.dataset) == ("data/mvtec_ad", ["dino_vitbase8"], ["blocks.10", "blocks.11"], 2048,
                                               4096, "outputs", 3, 1, 1, "unsupervised", "mvtec_ad")


def test_sweep_csv_layout_and_pickle_round_trip(tmp_path, oracle_pdist, golden):
    """Write artefacts in main.py's format from the reference's committed X, then run test.py's sweep."""
    main, test_mod = _examples()
    out = str(tmp_path / "outputs")
    g = {"toothbrush": golden("kat_toothbrush_unsupervised.npz"), "grid": golden("kat_grid_unsupervised.npz")}
    layers = ["blocks.10", "blocks.11"]
    for cat, gg in g.items():
        folder = main.artefact_dir(out + "/mvtec_ad/dino_vitbase8/unsupervised", layers, 2048, 4096, 2, 1)
        os.makedirs(folder, exist_ok=True)
        n = len(gg["anomaly"])
        torch.save((torch.full((n, 1, 784), 1.0 / 784), gg["X"]),
                   folder + "/matrix_alpha_X_%s_unsupervised.pickle" % cat)
        os.makedirs(out + "/mvtec_ad/info", exist_ok=True)
        info = [{"classname": [cat], "anomaly": [str(a)], "is_anomaly": torch.tensor([int(a != "good")]),
                 "image_name": ["x"], "image_path": ["y"]} for a in gg["anomaly"]]
        torch.save(info, out + "/mvtec_ad/info/info_%s.pickle" % cat)
    files = test_mod.sweep(["unsupervised"], [2], 2048, 4096, ["dino_vitbase8"], layers, 3, 1, output_dir=out,
                           objects=["toothbrush"], textures=["grid"])
    assert files[0].endswith("blocks.10_blocks.11_2048_4096_tau_result.csv")
    rows = list(csv.reader(open(files[0], encoding="gbk")))
    assert rows[0] == ["unsupervised"] and rows[1] == ["Category", "NMI", "ARI", "F1"]
    assert rows[2] == ["---"] * 4 and rows[3] == ["TAU=2"]
    assert [r[0] for r in rows[4:]] == ["toothbrush", "grid", "MVTec(object)", "MVTec(texture)"]
    np.testing.assert_allclose([float(v) for v in rows[4][1:]], g["toothbrush"]["expected"], atol=1e-12)
    np.testing.assert_allclose([float(v) for v in rows[5][1:]], g["grid"]["expected"], atol=1e-12)
    np.testing.assert_allclose([float(v) for v in rows[6][1:]], g["toothbrush"]["expected"], atol=1e-12)


def test_alpha_heatmaps(tmp_path):
    """Reference test.py:158-174: 28x28 alpha -> x8 nearest -> / max; one file per anomaly type."""
    import PIL.Image
    main, test_mod = _examples()
    a = np.full(784, 0.001, dtype=np.float32)
    a[5 * 28 + 9] = 0.2                                    # the maximum at patch (5, 9)
    a[0] = 0.1
    heat = test_mod.alpha_heatmap(a)
    assert heat.shape == (224, 224) and heat.dtype == np.uint8 and heat.max() == 255
    assert (heat[40:48, 72:80] == 255).all() and (heat[0:8, 0:8] == 127).all() and heat[8, 8] == 1
    info = [{"classname": ["grid"], "anomaly": [k], "image_path": ["synthetic://none.png"]} for k in
            ("bent", "bent", "good", "good", "thread")]
    alpha = torch.from_numpy(np.stack([a] * 5)).unsqueeze(1)
    files = test_mod.render_heatmaps("grid", alpha, info, str(tmp_path / "vis"))
    assert [os.path.basename(f) for f in files] == ["grid_bent.png", "grid_good.png", "grid_thread.png"]
    im = np.asarray(PIL.Image.open(files[0]))
    assert im.shape == (224, 3 * 224 + 16, 3) and (im[40:48, 2 * 224 + 16 + 72: 2 * 224 + 16 + 80] == 255).all()
    with pytest.raises(ValueError):
        test_mod.alpha_heatmap(np.ones(10))


def test_synthetic_dataset_shapes_and_histograms():
    from patchcore.datasets import synthetic
    sizes = {c: sum(n for _, n in h) for c, h in synthetic.TEST_HISTOGRAMS.items()}
    assert list(sizes.values()) == [83, 150, 132, 117, 78, 110, 124, 115, 167, 160, 117, 42, 100, 79, 151]
    assert sum(sizes.values()) == 1725 and sum(synthetic.TRAIN_SIZES.values()) == 3629
    ds = synthetic.SyntheticMVTecDataset(classname="toothbrush", split=synthetic.DatasetSplit.TEST)
    assert len(ds) == 42
    item = ds[0]
    assert set(item) == {"image", "mask", "classname", "anomaly", "is_anomaly", "image_name", "image_path"}
    assert item["image"].shape == (3, 224, 224) and item["mask"].shape == (1, 224, 224)
    assert item["anomaly"] == "defective" and item["is_anomaly"] == 1 and ds[41]["anomaly"] == "good"
    again = synthetic.SyntheticMVTecDataset(classname="toothbrush", split=synthetic.DatasetSplit.TEST)
    assert torch.equal(ds.images, again.images)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=1)))
    assert batch["anomaly"] == ["defective"] and batch["is_anomaly"].tolist() == [1]


def test_mvtec_folder_dataset(tmp_path):
    import PIL.Image
    from patchcore.datasets import mvtec
    root = tmp_path / "mv"
    rng = np.random.default_rng(0)
    for split, kinds in (("train", ["good"]), ("test", ["good", "crack"])):
        for kind in kinds:
            os.makedirs(root / "nut" / split / kind)
            for i in range(2):
                PIL.Image.fromarray(rng.integers(0, 255, (300, 320, 3), dtype=np.uint8)).save(
                    root / "nut" / split / kind / ("%03d.png" % i))
    os.makedirs(root / "nut" / "ground_truth" / "crack")
    for i in range(2):
        PIL.Image.fromarray(rng.integers(0, 2, (300, 320), dtype=np.uint8) * 255).save(
            root / "nut" / "ground_truth" / "crack" / ("%03d_mask.png" % i))
    ds = mvtec.MVTecDataset(str(root), "nut", split=mvtec.DatasetSplit.TEST)
    assert len(ds) == 4 and [ds.data_to_iterate[i][1] for i in range(4)] == ["crack", "crack", "good", "good"]
    item = ds[0]
    assert item["image"].shape == (3, 224, 224) and item["mask"].shape == (1, 224, 224)
    assert item["image_name"] == "nut/test/crack/000.png" and item["is_anomaly"] == 1
    assert abs(float(item["image"].mean())) < 1.0 and float(item["mask"].max()) <= 1.0
    assert float(ds[2]["mask"].abs().sum()) == 0.0
    with pytest.raises(FileNotFoundError):
        mvtec.MVTecDataset(str(tmp_path / "missing"), "nut")


def test_backbone_registry(monkeypatch, tmp_path):
    import patchcore.backbones as bb
    with pytest.raises(NotImplementedError):
        bb.load("resnet50")
    assert set(bb._DINO) == {"dino_deitsmall16", "dino_deitsmall8_300ep", "dino_vitbase16", "dino_vitbase8"}
    # no weights and no opt-in: never a silent random init
    monkeypatch.delenv("AC_BACKBONE_SEED", raising=False)
    monkeypatch.delenv("AC_DINO_WEIGHTS", raising=False)
    with pytest.raises(RuntimeError, match="AC_BACKBONE_SEED"):
        bb.load("dino_vitbase8")
    with pytest.raises(RuntimeError, match="AC_BACKBONE_SEED"):
        bb.load("wideresnet50")
    # AC_DINO_WEIGHTS set but the checkpoint is not there: raise even with the seed opt-in present
    monkeypatch.setenv("AC_BACKBONE_SEED", "0")
    monkeypatch.setenv("AC_DINO_WEIGHTS", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        bb.load("dino_vitbase8")
    # timm's ViT-B/8 (reference backbones.py:36): same transformer + a head that is never run; its state_dict (DINO's
    # key names + norm / head) loads strictly from $AC_TIMM_WEIGHTS/<timm model name>.pth
    import torch
    import patchcore.vision_transformer as vits
    monkeypatch.delenv("AC_DINO_WEIGHTS", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    torch.manual_seed(3)
    src = vits._vit(768, 12, 8, depth=12, num_classes=1000)
    keys = list(src.state_dict())
    assert keys[:4] == ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"]
    assert keys[-4:] == ["norm.weight", "norm.bias", "head.weight", "head.bias"] and len(keys) == 152
    torch.save(src.state_dict(), tmp_path / "vit_base_patch8_224.pth")
    monkeypatch.setenv("AC_TIMM_WEIGHTS", str(tmp_path))
    m = bb.load("vit_base")
    assert m.weights_source.endswith("vit_base_patch8_224.pth") and tuple(m.pos_embed.shape) == (1, 785, 768)
    assert torch.equal(m.blocks[11].mlp.fc2.weight, src.blocks[11].mlp.fc2.weight)
    assert set(bb._TIMM_VIT) == {"vit_small", "vit_base", "vit_large", "vit_deit_base"}


def test_shims_construct_and_refuse_detection_calls():
    import patchcore.common as common
    import patchcore.sampler
    nn = common.FaissNN(True, 4)
    scorer = common.NearestNeighbourScorer(n_nearest_neighbours=5, nn_method=nn)
    seg = common.RescaleSegmentor(device="cpu", target_size=(224, 224))
    assert scorer.nn_method is nn and seg.smoothing == 4
    assert patchcore.sampler.IdentitySampler().run(3) == 3
    with pytest.raises(NotImplementedError):
        scorer.fit([])
    pre = common.Preprocessing([785, 785], 2048)
    assert pre.output_dim == 2048 and len(pre.preprocessing_modules) == 2
    assert common.Aggregator(target_dim=4096).target_dim == 4096


def test_bench_workloads_match_baseline_configs():
    """bench.py --config: the workloads BASELINE.json / SURVEY 8d name (no GPU needed to build them)."""
    import argparse
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("ac_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cpu = torch.device("cpu")

    def build(config, **kw):
        ns = argparse.Namespace(config=config, category=None, n=kw.get("n", 40), tau=1.0)
        return bench.build_workload(ns, cpu)

    name, items, mode = build("2")
    assert mode == "unsupervised" and len(items) == 1 and items[0]["images"].shape == (83, 3, 224, 224) and "N=83" in name
    name, items, mode = build("2-pill")
    assert items[0]["category"] == "pill" and len(items[0]["labels"]) == 167
    name, items, mode = build("5a", n=40)
    assert mode == "average" and items[0]["images"].shape == (40, 3, 224, 224) and len(set(items[0]["labels"])) == 5
    assert items[0]["images"][0:40].shape == (40, 3, 224, 224)        # a ShardView: this rank's shard only
    with pytest.raises(IndexError):
        items[0]["images"][0:7]
    ns = argparse.Namespace(config="5b", category=None, n=10, tau=1.0)
    shard = bench.build_workload(ns, cpu, rank=1, world=4)[1][0]["images"]
    assert shard.shape[0] == 10 and shard[3:6].shape[0] == 3
    assert build("5b", n=12)[2] == "unsupervised"
    assert bench.CONFIGS == ("2", "2-pill", "3", "4", "5a", "5b")


def test_category_rotation_spreads_the_large_shards():
    """run_categories rotates the shards of category c by category_rot(c, world): with the 15 MVTec test-set sizes on 8
    ranks no rank holds the ceil(N / 8) shard of EVERY category any more (221 images on seven ranks and 178 on the last
    without rotation; 212-219 with it)."""
    from patchcore.pipeline import category_rot, shard_bounds
    sizes = [83, 150, 132, 117, 78, 110, 124, 115, 167, 160, 117, 42, 100, 79, 151]
    world = 8
    plain = [sum(shard_bounds(n, world)[r][1] - shard_bounds(n, world)[r][0] for n in sizes) for r in range(world)]
    rotated = [sum(shard_bounds(n, world)[(r + category_rot(c, world)) % world][1] - shard_bounds(n, world)[(r + category_rot(c, world)) % world][0]
                   for c, n in enumerate(sizes)) for r in range(world)]
    assert sum(plain) == sum(rotated) == 1725 and max(plain) == 221
    assert plain[:7] == [221] * 7 and plain[7] == 178            # seven ranks carry the ceil() shard of every category
    assert max(rotated) <= 219 and max(rotated) - min(rotated) <= 7
    assert category_rot(5, 1) == 0 and [category_rot(c, 3) for c in range(4)] == [0, 1, 2, 0]


def test_vit_batch_plan_fills_whole_tile_rounds():
    """Launch sizes of the ViT over a stream of images (pipeline.vit_batch_plan): every plan covers the stream, no launch
    exceeds the limit, the sizes used are those whose tile grid (785 tokens per image, 256-row tiles, 256 CUs) ends just
    below a whole round of the chip - 83, 111, 138, 166, 194, 222, 250 for ViT-B/8, measured 5 % cheaper per image than
    the next size up - and a short stream is one launch."""
    from patchcore.pipeline import vit_batch_cost, vit_batch_plan, _vit_rounds
    good = [27, 55, 83, 111, 138, 166, 194, 222, 250]
    for g in good:
        r, r1 = _vit_rounds(g, 785, 768, 256), _vit_rounds(g + 1, 785, 768, 256)
        assert r <= round(r) + 1e-9 and round(r) - r < 0.05 and r1 > round(r)        # g fills its rounds, g + 1 spills
        assert vit_batch_cost(g, 785, 768) / g < vit_batch_cost(g + 1, 785, 768) / (g + 1)
    for total, limit in [(1725, 128), (1725, 256), (3450, 256), (10000, 256), (167, 128), (129, 128), (84, 83), (5, 128), (300, 32)]:
        plan = vit_batch_plan(total, limit, 785, 768, 256)
        assert sum(plan) == total and all(0 < b <= limit for b in plan)
        if total <= limit:
            assert plan == [total]
        else:
            naive = [limit] * (total // limit) + ([total % limit] if total % limit else [])
            assert sum(vit_batch_cost(b, 785, 768) for b in plan) <= sum(vit_batch_cost(b, 785, 768) for b in naive) + 1e-9
            assert plan == naive or all(b in good for b in plan[:-1])                 # only the last launch is ragged
    assert vit_batch_plan(0, 128, 785, 768) == []
    # other widths / chips: ViT-S (one column tile of 384 -> 2), 304 CUs
    plan = vit_batch_plan(1000, 200, 197, 384, 304)
    assert sum(plan) == 1000 and max(plan) <= 200


def test_batch_edges_follow_a_plan_or_a_size():
    from patchcore.patchcore import _batch_edges
    assert _batch_edges(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert _batch_edges(10, [3, 3, 4]) == [(0, 3), (3, 6), (6, 10)]
    assert _batch_edges(0, 4) == [] and _batch_edges(5, 8) == [(0, 5)]
    with pytest.raises(ValueError):
        _batch_edges(10, [3, 3])                          # a plan must cover every image
    with pytest.raises(ValueError):
        _batch_edges(3, [3, 0])
