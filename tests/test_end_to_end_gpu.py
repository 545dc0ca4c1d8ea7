"""GPU: the drop-in scripts end to end (examples/main.py -> pickle -> examples/test.py) on a small
MVTec-shaped synthetic category, against the oracle run on the same images and weights:
alpha (relative <= 1e-4 at tau = 1, absolute <= 1e-6: BASELINE.md section 3's gate; alpha amplifies the
fp32-level differences of the ViT and of w by 1/tau), X (<= 1e-4 absolute) and NMI / ARI / F1 within +-1e-3
(north_star's bar).  The tau sweep is checked per tau against the oracle's alpha."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import EXAMPLES

pytestmark = pytest.mark.gpu


def _load_examples():
    import importlib.util
    mods = []
    for name in ("main", "test"):
        spec = importlib.util.spec_from_file_location("ac_examples_" + name, os.path.join(EXAMPLES, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods.append(mod)
    return mods


@pytest.fixture(scope="module")
def small_model(dev):
    """dino_vitbase8 geometry with 4 blocks (taps on the last two) to keep the CPU oracle fast."""
    import patchcore.vision_transformer as vits
    torch.manual_seed(0)
    m = vits.vit_base(patch_size=8, num_classes=0, depth=4)
    for p in m.parameters():
        p.requires_grad = False
    return m.eval().to(dev)


def _oracle_Z(model, images):
    from oracle import vit as ovit, aggregate as oagg
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    out = []
    for i in range(0, images.shape[0], 4):
        taps = ovit.forward_taps(sd, images[i:i + 4], taps=(2, 3))
        out.append(oagg.aggregate([taps[2], taps[3]], 3, 2048, 4096)[0].reshape(-1, 784, 4096))
    return torch.cat(out)


def _oracle_category(model, images, mode, tau, train_images=None, bank_idx=None):
    from oracle import vit as ovit, aggregate as oagg, alpha as oalpha
    sd = {k: v.cpu() for k, v in model.state_dict().items()}

    def embed(x):
        out = []
        for i in range(0, x.shape[0], 4):
            taps = ovit.forward_taps(sd, x[i:i + 4], taps=(2, 3))
            out.append(oagg.aggregate([taps[2], taps[3]], 3, 2048, 4096)[0].reshape(-1, 784, 4096))
        return torch.cat(out)

    Z = embed(images)
    if mode == "unsupervised":
        a64 = oalpha.matrix_alpha_unsupervised(tau, 1, Z)
    elif mode == "supervised":
        a64 = oalpha.matrix_alpha_supervised(tau, 1, Z, embed(train_images[bank_idx]))
    else:
        a64 = oalpha.matrix_alpha_average(Z)
    return oalpha.weighted_average(a64, Z)


@pytest.mark.parametrize("mode", ["unsupervised", "supervised", "average"])
def test_main_then_test_scripts_match_oracle(tmp_path, dev, small_model, mode, monkeypatch):
    main, test_mod = _load_examples()
    from patchcore.datasets import synthetic
    from patchcore.pipeline import select_bank
    from oracle import cluster as ocluster

    limit = 3                                                 # 3 images per anomaly type of "grid": N = 18, 6 classes
    orig = synthetic.SyntheticMVTecDataset

    def small(**kw):
        kw["limit_per_class"] = limit if kw.get("split") == synthetic.DatasetSplit.TEST else None
        ds = orig(**kw)
        if kw.get("split") != synthetic.DatasetSplit.TEST:     # shrink the train split too
            ds.anomalies, ds.images = ds.anomalies[:24], ds.images[:24]
        return ds
    monkeypatch.setattr(synthetic, "SyntheticMVTecDataset", small)

    out = str(tmp_path / "outputs")
    layers = ["blocks.2", "blocks.3"]
    save_path = out + "/synthetic/dino_vitbase8/" + mode
    alpha, X = main.make_category_data(
        path="nowhere", category="grid", pretrain_embed_dimension=2048, target_embed_dimension=4096,
        backbone_names=["dino_vitbase8"], layers_to_extract_from=layers, patchsize=3, save_path=save_path,
        tau=1, train_ratio=1.0, supervised=mode, dataset="synthetic", info_dir=out + "/synthetic/info",
        backbone=small_model, batch_size=8)
    n = X.shape[0]
    assert alpha.shape == (n, 1, 784) and alpha.dtype == torch.float32 and X.dtype == np.float32 and X.shape == (n, 4096)

    # the same inputs through the oracle
    test_ds = small(classname="grid", split=synthetic.DatasetSplit.TEST)
    train_ds = small(classname="grid", split=synthetic.DatasetSplit.TRAIN)
    bank = select_bank(len(train_ds), n, 1.0, 0)
    a_ref, X_ref = _oracle_category(small_model, test_ds.images, mode, 1.0, train_ds.images, bank)
    a = alpha.numpy()
    print(mode, "alpha max rel err %.2e  X max abs err %.2e"
          % (np.max(np.abs(a - a_ref.numpy()) / np.maximum(a_ref.numpy(), 1e-12)), np.abs(X - X_ref).max()))
    # alpha rel <= 1e-4 (tau >= 0.2), abs <= 1e-6.  Budget at tau = 1: d(alpha)/alpha = dw
    # (both sides), dw = |w_hip - w_oracle| <= 2e-5 (each side's distance from the fp64 truth, test_alpha_gpu.py) +
    # the ViT's 1.5e-5 on the features feeding it; measured 1.4e-5 .. 2.7e-5.
    np.testing.assert_allclose(a, a_ref.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(X, X_ref, rtol=0, atol=1e-4)

    # consumer script on the artefacts it just wrote vs the oracle's clustering of the oracle's X
    NMI, ARI, F1, label, predict = test_mod.calculate_metrics(
        category="grid", pretrain_embed_dimension=2048, target_embed_dimension=4096,
        backbone_names=["dino_vitbase8"], layers_to_extract_from=layers, patchsize=3, train_ratio=1.0, tau=1,
        supervised=mode, dataset="synthetic", output_dir=out)
    rN, rA, rF, rl, rp = ocluster.calculate_metrics(X_ref, list(test_ds.anomalies))
    print(mode, "NMI/ARI/F1 ours %.4f %.4f %.4f  oracle %.4f %.4f %.4f" % (NMI, ARI, F1, rN, rA, rF))
    assert abs(NMI - rN) <= 1e-3 and abs(ARI - rA) <= 1e-3 and abs(F1 - rF) <= 1e-3
    assert np.array_equal(label, rl)


def test_tau_sweep_reuses_one_pass(dev, small_model):
    """run_category with a list of taus: one embedding + distance pass; every tau equals its own single run AND
    the oracle's alpha for that tau on the same images (reference: utils.py:240-257 re-run per tau)."""
    import patchcore.common as common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from patchcore.pipeline import HipOps, run_category
    from patchcore.datasets import synthetic
    from oracle import alpha as oalpha
    core = AnomalyClusteringCore(dev)
    core.load(backbone=small_model, layers_to_extract_from=["blocks.2", "blocks.3"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), nn_method=common.FaissNN(False, 4))
    images = synthetic.make_images("grid", ["good", "bent", "bent", "glue", "good"])
    ops = HipOps(core, batch_size=8)
    taus = [0, 0.2, 0.5, 2.0, 20.0]
    sweep = run_category(ops, images.to(dev), "unsupervised", tau=taus)
    assert sorted(sweep["by_tau"]) == sorted(float(t) for t in taus)
    Z_ref = _oracle_Z(small_model, images)
    w_ref = oalpha.weights_unsupervised(Z_ref)
    for tv in taus:
        single = run_category(ops, images.to(dev), "unsupervised", tau=tv)
        got = sweep["by_tau"][float(tv)]
        assert torch.equal(single["alpha"], got["alpha"])
        assert np.array_equal(single["X"], got["X"])
        a_ref = torch.cat([oalpha.alpha_from_weights(w, tv, 1) for w in w_ref], 0).float().numpy()
        a = got["alpha"].squeeze(1).cpu().numpy()
        if tv == 0:
            assert np.array_equal(a > 0, a_ref > 0), "tau = 0: arg-max indicator differs from the oracle"
        else:
            # d(alpha)/alpha = dw / tau: the 1e-4 bar of tau = 1 scales with 1/tau below it (tau = 0.2: 5e-4)
            np.testing.assert_allclose(a, a_ref, rtol=1e-4 / min(1.0, tv), atol=1e-6, err_msg="tau=%g" % tv)
    a0 = sweep["by_tau"][0.0]["alpha"].squeeze(1)
    assert torch.all((a0 > 0).sum(1) >= 1) and torch.allclose(a0.sum(1), torch.ones(5, device=dev))


def test_cli_both_modes_share_one_embedding_pass(tmp_path, dev, monkeypatch):
    """`main.py --supervised both --dataset synthetic --categories toothbrush`: the reference's hard-coded loop over
    both modes (main.py:348).  The test set is loaded and embedded once per category and reused by the second
    mode; both artefacts equal what separate single-mode runs write, the info pickle is written, and the run is
    marked as a random-init run."""
    main, test_mod = _load_examples()
    out = str(tmp_path / "o")
    argv = ["--dataset", "synthetic", "--categories", "toothbrush", "--output_dir", out, "--batch_size", "64"]
    both = main.main(argv + ["--supervised", "both"])
    assert set(both) == {("unsupervised", "toothbrush"), ("supervised", "toothbrush")}
    for mode in ("unsupervised", "supervised"):
        single = main.main(argv + ["--supervised", mode, "--output_dir", out + "_" + mode])
        a, X = both[(mode, "toothbrush")]
        a1, X1 = single[(mode, "toothbrush")]
        assert a.shape == (42, 1, 784) and torch.equal(a, a1) and np.array_equal(X, X1)
        folder = out + "/synthetic/dino_vitbase8/" + mode
        assert open(folder + "/WEIGHTS_SOURCE.txt").read().strip() == "random-init(seed=0)"
        saved = test_mod._safe_load(folder + "/blocks.10_blocks.11_2048_4096_1.0_1.0/matrix_alpha_X_toothbrush_%s.pickle" % mode)
        assert torch.equal(saved[0], a) and np.array_equal(saved[1], X)
    info = test_mod._safe_load(out + "/synthetic/info/info_toothbrush.pickle")
    assert len(info) == 42 and info[0]["anomaly"] == ["defective"] and info[41]["anomaly"] == ["good"]
    with pytest.raises(FileNotFoundError):                       # a mistyped --path never silently goes synthetic
        main.main(["--path", str(tmp_path / "nope"), "--categories", "toothbrush", "--output_dir", out])


def test_cli_categories_as_one_job_write_the_same_artefacts(tmp_path, dev):
    """`main.py --dataset synthetic --categories toothbrush grid --supervised both` three ways: the categories as ONE
    job (the default: joint ViT batches over both categories, pipeline.run_categories), one after another (the
    reference's loop, examples/main.py:346-366) and whole categories per GPU: the same pickles.  A ViT batch that spans
    the category boundary gives every image the embedding its own batch gives it (the GEMM kernels accumulate each
    output row in one fixed K order), so the files are equal to the bit."""
    main, test_mod = _load_examples()
    argv = ["--dataset", "synthetic", "--categories", "toothbrush", "grid", "--supervised", "both", "--batch_size", "64"]
    res = {}
    for sched in ("pipeline", "sequential", "per_gpu"):
        res[sched] = main.main(argv + ["--output_dir", str(tmp_path / sched), "--category_schedule", sched])
    keys = {(m, c) for m in ("unsupervised", "supervised") for c in ("toothbrush", "grid")}
    assert set(res["pipeline"]) == keys and set(res["sequential"]) == keys and set(res["per_gpu"]) == keys
    for key in keys:
        a, X = res["sequential"][key]
        for sched in ("pipeline", "per_gpu"):
            a1, X1 = res[sched][key]
            assert torch.equal(a, a1) and np.array_equal(X, X1), (sched, key)
        mode, cat = key
        f = "/synthetic/dino_vitbase8/%s/blocks.10_blocks.11_2048_4096_1.0_1.0/matrix_alpha_X_%s_%s.pickle" % (mode, cat, mode)
        saved = test_mod._safe_load(str(tmp_path / "pipeline") + f)
        assert torch.equal(saved[0], a) and np.array_equal(saved[1], X)
    assert len(test_mod._safe_load(str(tmp_path / "pipeline") + "/synthetic/info/info_grid.pickle")) == 78


@pytest.mark.parametrize("schedule", ["pipeline", "sequential"])
def test_cli_under_torchrun_writes_the_one_process_artefacts(tmp_path, dev, schedule):
    """`python -m torch.distributed.run --nproc-per-node 2 main.py ...` - how the reference's CLI runs on several GPUs -
    rehearsed on this box's one GPU (--rehearse_on_one_gpu: the ranks share cuda:0 over gloo).  Every rank decodes only
    its own image shard, the ring carries the ViT taps, rank 0 writes the pickles: equal, bit for bit, to those of the
    plain one-process run."""
    import socket
    import subprocess
    import sys
    main, test_mod = _load_examples()
    argv = ["--dataset", "synthetic", "--categories", "toothbrush", "grid", "--supervised", "both", "--batch_size", "64",
            "--category_schedule", schedule]
    ref = main.main(argv + ["--output_dir", str(tmp_path / "one")])
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = os.path.join(EXAMPLES, "main.py")
    env = dict(os.environ, AC_BACKBONE_SEED=os.environ.get("AC_BACKBONE_SEED", "0"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), script] + argv +
                       ["--output_dir", str(tmp_path / "two"), "--rehearse_on_one_gpu"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    for (mode, cat), (a, X) in ref.items():
        f = "/synthetic/dino_vitbase8/%s/blocks.10_blocks.11_2048_4096_1.0_1.0/matrix_alpha_X_%s_%s.pickle" % (mode, cat, mode)
        saved = test_mod._safe_load(str(tmp_path / "two") + f)
        assert torch.equal(saved[0], a) and np.array_equal(saved[1], X), (mode, cat)
    assert len(test_mod._safe_load(str(tmp_path / "two") + "/synthetic/info/info_grid.pickle")) == 78
