#!/usr/bin/env python3
"""Per-rank COMPUTE time of the image-sharded schedule at G = 1, 2, 4, 8 ranks, measured on ONE MI355X.

Rank 0 of G (a full shard: the slowest rank) embeds its ceil(N/G) images, packs them and walks its own part of the
unsupervised block schedule (pipeline.unsup_schedule: the symmetric block of its shard + the two-way blocks against
the G/2 shards it receives) - with the "received" shards taken from a local packed copy of the whole category, so no
communication happens.  What the 8-GPU run adds on top is listed per G as bytes: the ring transfers (posted up front,
under the step-0 work), the int64 all-reduce of the sums and the all-gather of X / alpha.  Output: one JSON object
(profiles/r03_scaling_prediction.json is a committed run) - the prediction DESIGN.md section 6 quotes, for the first
multi-GPU measurement to be judged against.

    python scratch/predict_scaling.py [--categories bottle pill] [--synthetic-n 2000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "anomaly-clustering_amd", "models")]
os.environ.setdefault("AC_BACKBONE_SEED", "0")

import torch  # noqa: E402

P, D = 784, 4096
XGMI_LINK_GBS = 50.0      # assumed EFFECTIVE one-direction rate of one xGMI link for a large P2P copy (spec ~64 raw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--categories", nargs="+", default=["bottle", "pill"])
    ap.add_argument("--all15", action="store_true")
    ap.add_argument("--synthetic-n", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import acnative
    import patchcore.backbones
    import patchcore.common as common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from patchcore.pipeline import HipOps, shard_bounds, unsup_schedule, block_images, MEAN, NO_SELF
    from patchcore.datasets import synthetic
    dev = torch.device("cuda:0")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=patchcore.backbones.load("dino_vitbase8"), layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    ops = HipOps(core, batch_size=128)
    cats = list(synthetic.TEST_HISTOGRAMS) if a.all15 else a.categories
    work = [(c, synthetic.category_labels(c)) for c in cats]
    if a.synthetic_n:
        kinds = ["good", "a", "b", "c", "d"]
        work.append(("synthetic", [kinds[(i * 5) // a.synthetic_n] for i in range(a.synthetic_n)]))

    def sync_time(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    result = {"assumed_xgmi_link_GBs": XGMI_LINK_GBS, "categories": []}
    for cat, labels in work:
        n = len(labels)
        images = synthetic.make_images(cat, labels)
        Z = ops.embed(images.to(dev) if n <= 512 else images)
        pk_all = ops.pack(Z)
        shift = ops.fixed_shift(ops.max_sq([pk_all], dev), n)
        row = {"category": cat, "n": n, "per_G": {}}
        for G in (1, 2, 4, 8):
            bounds = shard_bounds(n, G)
            lo, hi = bounds[0]
            c = block_images(P)
            sched = list(unsup_schedule(bounds, 0, c))
            mine = images[lo:hi].to(dev)

            def rank_step():
                t_e, Zl = sync_time(lambda: ops.embed(mine))
                def alpha():
                    pk = ops.pack(Zl)
                    del pk
                    acc = ops.new_acc(n * P, MEAN, dev)
                    for it in sched:
                        if it[0] == "sym":
                            A = pk_all.slice(*it[2])
                            ops.pair_accumulate(ops.pair_block(A, A), P, MEAN, 0, shift, acc[it[2][0] * P: it[2][1] * P])
                        else:
                            x, y = it[2], it[3]
                            if x[0] > y[0]:
                                x, y = y, x
                            mb, mbt = ops.pair_block(pk_all.slice(*x), pk_all.slice(*y), two_way=True)
                            ops.pair_accumulate(mb, P, MEAN, NO_SELF, shift, acc[y[0] * P: y[1] * P])
                            ops.pair_accumulate(mbt, P, MEAN, NO_SELF, shift, acc[x[0] * P: x[1] * P])
                    w = ops.pair_finalize(acc[lo * P: hi * P], P, MEAN, n - 1, shift)
                    al = ops.alpha_softmax(w, 1.0, 1).float().contiguous()
                    return ops.weighted_avg(al, Zl)
                t_a, _ = sync_time(alpha)
                return t_e, t_a
            rank_step()
            ts = [rank_step() for _ in range(a.reps)]
            te, ta = min(t[0] for t in ts), min(t[1] for t in ts)
            shard_bytes = (hi - lo) * P * (D * 4 + 4)
            steps = G // 2
            row["per_G"][G] = {
                "images_per_rank": hi - lo, "block_launches": len(sched), "embed_ms": 1e3 * te, "alpha_ms": 1e3 * ta,
                "compute_ms": 1e3 * (te + ta),
                "ring_shards_received": steps, "ring_bytes_per_shard": shard_bytes if G > 1 else 0,
                "ring_ms_per_shard_at_assumed_rate": 1e3 * shard_bytes / (XGMI_LINK_GBS * 1e9) if G > 1 else 0.0,
                "allreduce_bytes": n * P * 8 if G > 1 else 0, "allgather_bytes": n * (D * 4 + P * 4) if G > 1 else 0,
                "images_per_sec_if_comm_hidden": n / (te + ta)}
        base = row["per_G"][1]["compute_ms"]
        for G in (2, 4, 8):
            row["per_G"][G]["speedup_if_comm_hidden"] = base / row["per_G"][G]["compute_ms"]
        result["categories"].append(row)
        del Z, pk_all
        acnative.release_workspaces()
        torch.cuda.empty_cache()
    if len(result["categories"]) > 1:
        tot = {G: sum(r["per_G"][G]["compute_ms"] for r in result["categories"]) for G in (1, 2, 4, 8)}
        result["sum_over_categories_compute_ms"] = tot
        result["sum_speedup_if_comm_hidden"] = {G: tot[1] / tot[G] for G in (2, 4, 8)}
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
