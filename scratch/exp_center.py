"""Experiment: w error vs fp64 truth, single vs two-level accumulation, with/without centring."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anomaly-clustering_amd", "models")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import acnative
dev = torch.device("cuda:0")

def truth_rows(Z, idx):
    """fp64 w for images idx (unsupervised mean over others)."""
    N, P, D = Z.shape
    Zd = Z.double()
    sq = (Zd * Zd).sum(2)
    out = []
    for i in idx:
        acc = torch.zeros(P, dtype=torch.float64, device=Z.device)
        for j in range(N):
            if j == i: continue
            d2 = sq[i][:, None] + sq[j][None, :] - 2 * Zd[i] @ Zd[j].T
            acc += d2.clamp_min(0).sqrt().min(1)[0]
        out.append(acc / (N - 1))
    return torch.stack(out)

def run(Z, tag, idx):
    N, P, D = Z.shape
    flat = Z.reshape(-1, D)
    c = flat.double().mean(0)
    Zc = (Z - c.float()).contiguous()
    st = {"tag": tag, "gemm": os.environ.get("AC_GEMM", "default"),
          "mean_sq": float((flat.double() ** 2).sum(1).mean()), "mean_sq_centred": float((Zc.reshape(-1, D).double() ** 2).sum(1).mean()),
          "max_sq_centred": float((Zc.reshape(-1, D).double() ** 2).sum(1).max())}
    t = truth_rows(Z, idx)
    st["w_mean"] = float(t.mean()); st["w_min"] = float(t.min())
    for name, X in (("plain", Z), ("centred", Zc)):
        acnative.pair_min_dist_staged(X, X, 0)
        torch.cuda.synchronize()
        acnative.kernel_events = []
        w = acnative.pair_min_dist_staged(X, X, 0)
        torch.cuda.synchronize()
        ms = acnative.kernel_events[0][1].elapsed_time(acnative.kernel_events[0][2]); acnative.kernel_events = None
        st[name + "_err"] = float((w[idx].double() - t).abs().max())
        st[name + "_ms"] = ms
    print(json.dumps(st), flush=True)

which = sys.argv[1]
if which == "synth":
    from _synth import clustered_embeddings
    Zf, _ = clustered_embeddings(3, 784, 4096, 3, salt=41)
    run(torch.from_numpy(Zf).to(dev), "alpha_full", [0, 1, 2])
else:
    import patchcore.backbones, patchcore.common as common, patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from patchcore.pipeline import HipOps
    from patchcore.datasets import synthetic
    backbone = patchcore.backbones.load("dino_vitbase8")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=backbone, layers_to_extract_from=["blocks.10", "blocks.11"], device=dev, input_shape=(3, 224, 224),
              pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    labels = synthetic.category_labels("bottle")
    images = synthetic.make_images("bottle", labels).to(dev)
    Z = HipOps(core, batch_size=128).embed(images)
    run(Z, "bench_bottle", [0, 41, 82])
