#!/usr/bin/env python3
"""Config 5b at its FULL size on one GPU, checked: N = 10 000 synthetic images, unsupervised alpha (lean mode: the rank keeps
the ViT taps instead of Z), then
  * every ordered image pair resolved exactly once (the block launches' pair counts sum to N (N - 1)),
  * alpha finite, positive, rows summing to 1,
  * w of three images (first block, a middle block, the last block) against the float64 brute force over all 9 999 other
    images (Z rebuilt from the taps chunk by chunk), gate 2e-5 as in tests/test_configs_full_gpu.py.
Takes ~10 minutes of GPU time; prints a progress line a minute.  Writes gpurun_out/verify_5b_n10000.json (committed per round as profiles/rNN_verify_5b_n10000.json).

    python scratch/verify_5b_n10000.py [N]"""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "anomaly-clustering_amd", "models")]
os.environ.setdefault("AC_BACKBONE_SEED", "0")
import acnative  # noqa: E402
import patchcore.backbones  # noqa: E402
import patchcore.common as common  # noqa: E402
import patchcore.sampler  # noqa: E402
from patchcore.datasets import synthetic  # noqa: E402
from patchcore.patchcore import AnomalyClusteringCore  # noqa: E402
from patchcore.pipeline import HipOps, Taps, run_category  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    dev = torch.device("cuda:0")
    t_start = time.time()

    def beat():
        while True:
            time.sleep(60)
            free, total = torch.cuda.mem_get_info(dev)
            print("verify: %d s, %.1f GB of HBM in use" % (time.time() - t_start, (total - free) / 1e9), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()

    core = AnomalyClusteringCore(dev)
    core.load(backbone=patchcore.backbones.load("dino_vitbase8"), layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    kinds = ["good", "a", "b", "c", "d"]
    labels = [kinds[(i * len(kinds)) // n] for i in range(n)]
    images = synthetic.make_images("synthetic", labels)                 # host; embedded batch by batch
    ops = HipOps(core, batch_size=256)
    acnative.kernel_events = []
    t0 = time.time()
    out = run_category(ops, images, "unsupervised", tau=1.0, want_host=False, lean=True)
    torch.cuda.synchronize()
    step_s = time.time() - t0
    events, acnative.kernel_events = acnative.kernel_events, None
    pairs = sum(e[3] for e in events)
    taps = out["Z_loc"]
    assert isinstance(taps, Taps) and taps.n == n
    alpha = out["alpha"].squeeze(1)
    P = alpha.shape[1]
    res = {"n": n, "step_s": step_s, "launches": len(events), "ordered_pairs": pairs, "ordered_pairs_expected": n * (n - 1.0),
           "alpha_finite": bool(torch.isfinite(alpha).all()), "alpha_positive": bool((alpha > 0).all()),
           "alpha_row_sum_max_dev": float((alpha.double().sum(1) - 1.0).abs().max())}
    print("step %.1f s, %d launches, pairs %.0f" % (step_s, len(events), pairs), file=sys.stderr, flush=True)

    # float64 brute force for three images against all the others; key images rebuilt from the taps 48 at a time
    idx = [0, n // 2 - 1, n - 1]
    q = torch.cat([core.aggregate_taps([f[i:i + 1] for f in taps.feats]) for i in idx]).double()      # [3, P, D]
    qsq = (q * q).sum(2)
    tot = torch.zeros((len(idx), P), dtype=torch.float64, device=dev)
    for j0 in range(0, n, 48):
        k = core.aggregate_taps([f[j0:j0 + 48] for f in taps.feats]).double()
        ksq = (k * k).sum(2)
        for a, i in enumerate(idx):
            d2 = qsq[a][None, :, None] + ksq[:, None, :] - 2.0 * torch.matmul(q[a][None], k.transpose(1, 2))
            dmin = d2.clamp_min(0).sqrt().min(2)[0]                                   # [c, P]
            if j0 <= i < j0 + k.shape[0]:
                dmin[i - j0] = 0.0                                                     # the image itself is not a key
            tot[a] += dmin.sum(0)
    truth = tot / (n - 1)
    err = (out["w_loc"][idx].double() - truth).abs().max().item()
    res.update({"w_checked_images": idx, "w_max_abs_err_vs_fp64": err, "w_mean": float(truth.mean()), "w_gate": 2e-5})
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "verify_5b_n10000.json"), "w"), indent=1)
    ok = (abs(pairs - n * (n - 1.0)) < 1e-3 and res["alpha_finite"] and res["alpha_positive"] and res["alpha_row_sum_max_dev"] <= 1e-5
          and err <= 2e-5)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
