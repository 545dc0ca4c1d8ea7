#!/usr/bin/env python3
"""Per-rank COMPUTE time of the multi-GPU schedules at G = 1, 2, 4, 8 ranks, MEASURED on ONE MI355X, plus a timeline
of the ring transfers at an ASSUMED link rate - the prediction DESIGN.md section 6 quotes, for the first multi-GPU
measurement to be judged against.

Rank 0 of G (a full shard of every category: the slowest rank) runs exactly the code of pipeline.run_categories /
run_category - the same jobs (_UnsupJob), the same block launches in the same order - with the "received" shards taken
from a local packed copy of the whole category, so no communication happens:

  sequential   per category: embed the shard (its own ViT batches), pack, ring steps 0 .. G/2 (round 3's schedule)
  pipeline     embed the shards of ALL categories as one stream of full ViT batches, pack all, step 0 of all
               categories, then steps 1 .. G/2 category by category (run_categories' order)
  per_gpu      whole categories to ranks by longest-processing-time-first: the G = 1 per-category times, binned

MEASURED: every *_ms below that is not marked est.  ASSUMED: the xGMI rate (--link-gbs, default 50 GB/s effective
per link and direction); everything derived from it is marked est.  Re-measure the rate on the first multi-GPU lease
with `python scratch/p2p_probe.py` (hipMemcpyPeerAsync between two devices) and re-run this script with --link-gbs.

    python scratch/predict_scaling.py --all15            # config 3
    python scratch/predict_scaling.py --categories bottle
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


class LocalRing:
    """Stands in for pipeline._Ring: the shard of rank + k is a slice of a local copy of the whole category - of its
    TAPS (what the ring carries since round 4), which get() turns into a packed block exactly as the real ring does
    (aggregate + pack: part of the measured compute), or of its packed embeddings (--payload packed)."""

    contig = True          # like a ring whose steps were all posted up front: a run of steps is one block (get_run)
    pending = ()

    def __init__(self, ops, src, bounds, taps, rank=0):
        self.ops, self.src, self.bounds, self.taps, self.rank = ops, src, bounds, taps, rank

    def get(self, k):
        return self.get_run(k, k)

    def get_run(self, k1, k2):
        """Steps k1 .. k2 never wrap round image 0 inside one run (merge_later_steps), so they are one slice."""
        G = len(self.bounds)
        lo, hi = self.bounds[(self.rank + k1) % G][0], self.bounds[(self.rank + k2) % G][1]
        if hi <= lo:
            return None
        blk = self.src.slice(lo, hi)
        return self.ops.taps_to_packed(blk) if self.taps else blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--categories", nargs="+", default=["bottle"])
    ap.add_argument("--all15", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--link-gbs", type=float, default=50.0, help="ASSUMED effective one-direction rate of one xGMI link")
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--payload", default="taps", choices=["taps", "packed"], help="what the ring carries")
    a = ap.parse_args()
    import acnative
    import patchcore.backbones
    import patchcore.common as common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from patchcore.pipeline import HipOps, shard_bounds, _UnsupJob, lpt_assignment, category_rot
    from patchcore.datasets import synthetic
    dev = torch.device("cuda:0")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=patchcore.backbones.load("dino_vitbase8"), layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    ops = HipOps(core, batch_size=128)
    cats = list(synthetic.TEST_HISTOGRAMS) if a.all15 else a.categories
    labels = {c: synthetic.category_labels(c) for c in cats}
    ns = {c: len(labels[c]) for c in cats}
    images = {c: synthetic.make_images(c, labels[c]).to(dev) for c in cats}
    # the whole category packed once: the source of the "received" shards
    pk_all, src_all = {}, {}
    taps_mode = a.payload == "taps"
    for c in cats:
        Zc, tc = ops.embed(images[c], want_taps=True)
        pk_all[c] = ops.pack(Zc)
        src_all[c] = ops.taps_with_norms(tc, pk_all[c]) if taps_mode else pk_all[c]     # (what a real ring carries: taps + norms)
        del Zc
    per_image_bytes = src_all[cats[0]].bytes_per_image() if taps_mode else P * (D * 4 + 4)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0), out

    def jobs_for(G, Z):
        """One _UnsupJob per category for rank 0 of G (virtual rank category_rot(c, G): run_categories rotates the shards),
        prepared, with the local stand-in ring."""
        jobs = {}
        for ci, c in enumerate(cats):
            bounds = shard_bounds(ns[c], G)
            v = category_rot(ci, G)
            job = _UnsupJob(ops, ops.pack(Z[c]), bounds, v, G, None, dev)
            job.m = ops.max_sq([pk_all[c]], dev)                 # what the MAX all-reduce would deliver
            job.prepare()
            job.ring = LocalRing(ops, src_all[c], bounds, taps_mode, v)
            jobs[c] = job
        return jobs

    def tail(jobs, Z):
        for c in cats:
            w = jobs[c].finish()
            _, a32 = ops.alpha_softmax(w, 1.0, 1)
            ops.weighted_avg(a32, Z[c])

    result = {"ring_payload": a.payload, "ring_bytes_per_image": per_image_bytes, "assumed_xgmi_link_GBs": a.link_gbs, "categories": cats, "n": [ns[c] for c in cats], "per_G": {}}
    per_cat_1gpu = None
    for G in a.gpus:
        shards = {c: images[c][shard_bounds(ns[c], G)[0][0]: shard_bounds(ns[c], G)[0][1]] for c in cats}   # sequential: rank 0
        # pipeline: rank 0 holds the shard of virtual rank category_rot(c, G)
        shards_p = {c: images[c][shard_bounds(ns[c], G)[category_rot(ci, G)][0]: shard_bounds(ns[c], G)[category_rot(ci, G)][1]]
                    for ci, c in enumerate(cats)}
        steps = G // 2

        def run_sequential():
            t_embed = t_alpha = 0.0
            per_cat = []
            for c in cats:
                te, z = timed(lambda: ops.embed(shards[c]))
                def alpha():
                    jobs = jobs_for_one(c, z)
                    jobs.run(0)
                    jobs.run_later()
                    w = jobs.finish()
                    _, a32 = ops.alpha_softmax(w, 1.0, 1)
                    ops.weighted_avg(a32, z)
                ta, _ = timed(alpha)
                t_embed += te
                t_alpha += ta
                per_cat.append(te + ta)
            return t_embed, t_alpha, per_cat

        def jobs_for_one(c, z):
            bounds = shard_bounds(ns[c], G)
            job = _UnsupJob(ops, ops.pack(z), bounds, 0, G, None, dev)
            job.m = ops.max_sq([pk_all[c]], dev)
            job.prepare()
            job.ring = LocalRing(ops, src_all[c], bounds, taps_mode)
            return job

        def run_pipeline():
            te, Zs = timed(lambda: ops.embed_many([shards_p[c] for c in cats]))
            Z = dict(zip(cats, Zs))
            tp, jobs = timed(lambda: jobs_for(G, Z))                      # pack + norms + accumulators
            t0s, _ = timed(lambda: [jobs[c].run(0) for c in cats])        # step 0 of all categories
            later = []
            for c in cats:                                                # steps >= 1, category by category
                tc, _ = timed(lambda: jobs[c].run_later())
                later.append(tc)
            tt, _ = timed(lambda: tail(jobs, Z))
            return te, tp, t0s, later, tt

        run_sequential()
        seq = [run_sequential() for _ in range(a.reps)]
        best_seq = min(seq, key=lambda r: r[0] + r[1])
        run_pipeline()
        pipe = [run_pipeline() for _ in range(a.reps)]
        bp = min(pipe, key=lambda r: r[0] + r[1] + r[2] + sum(r[3]) + r[4])
        te, tp, t0s, later, tt = bp
        if G == 1:
            per_cat_1gpu = best_seq[2]
        # ---- ring timeline, est.: category c's `steps` shards travel concurrently (one link each), categories queue
        shard_mb = [shard_bounds(ns[c], G)[0][1] * per_image_bytes / 1e6 for c in cats]       # the full shard (a sender's)
        ms_per_cat = [mb / (a.link_gbs * 1e3) * 1e3 / 1.0 if G > 1 else 0.0 for mb in shard_mb]   # MB / (GB/s) = ms
        # sequential: each category's transfer starts after its own embed + pack and hides under its own step 0 only;
        # take step 0 of a category = its share of t0s (measured jointly), proportional to n^2
        sq = [ns[c] ** 2 for c in cats]
        step0 = [t0s * v / sum(sq) for v in sq]
        exposed_seq = sum(max(0.0, ms_per_cat[i] - step0[i]) for i in range(len(cats))) if G > 1 else 0.0
        # pipeline: all transfers posted when packing is done; category i has arrived after sum(ms_per_cat[:i+1]);
        # the compute reaches category i's step 1 at t0s + sum(later[:i])
        exposed_pipe, clock, arrived = 0.0, t0s, 0.0
        for i in range(len(cats)):
            arrived += ms_per_cat[i]
            if G > 1 and arrived > clock:
                exposed_pipe += arrived - clock
                clock = arrived
            clock += later[i]
        coll = 0.3 * len(cats) if G > 1 else 0.0                          # est.: 4 small collectives per category
        coll_pipe = 0.1 + 0.15 * len(cats) if G > 1 else 0.0              # est.: 2 per group + 2 gathers per category
        n_total = sum(ns.values())
        seq_ms = best_seq[0] + best_seq[1]
        pipe_ms = te + tp + t0s + sum(later) + tt
        result["per_G"][G] = {
            "images_on_rank0": sum(int(s.shape[0]) for s in shards.values()),
            "images_on_rank0_rotated": sum(int(s.shape[0]) for s in shards_p.values()),
            "sequential": {"embed_ms": best_seq[0], "alpha_ms": best_seq[1], "compute_ms": seq_ms,
                           "exposed_ring_ms_est": exposed_seq, "collectives_ms_est": coll,
                           "step_ms_est": seq_ms + exposed_seq + coll,
                           "images_per_sec_est": n_total / (seq_ms + exposed_seq + coll) * 1e3},
            "pipeline": {"embed_ms": te, "pack_ms": tp, "step0_all_ms": t0s, "later_steps_ms": sum(later), "tail_ms": tt,
                         "compute_ms": pipe_ms, "ring_mb_per_link_total": sum(shard_mb) if G > 1 else 0.0,
                         "ring_ms_total_est": sum(ms_per_cat), "exposed_ring_ms_est": exposed_pipe,
                         "collectives_ms_est": coll_pipe, "step_ms_est": pipe_ms + exposed_pipe + coll_pipe,
                         "images_per_sec_est": n_total / (pipe_ms + exposed_pipe + coll_pipe) * 1e3},
        }
        if per_cat_1gpu is not None and len(cats) > 1:
            table, loads = lpt_assignment(per_cat_1gpu, G)
            result["per_G"][G]["per_gpu"] = {"max_rank_ms": max(loads), "assignment": table,
                                             "images_per_sec_est": n_total / max(loads) * 1e3,
                                             "note": "G = 1 per-category times (measured, sequential) binned by LPT; "
                                                     "no communication but one all-gather of X / alpha"}
        acnative.release_workspaces()
        torch.cuda.empty_cache()
    base_seq = result["per_G"][a.gpus[0]]["sequential"]["step_ms_est"]
    base_pipe = result["per_G"][a.gpus[0]]["pipeline"]["step_ms_est"]
    for G in a.gpus:
        r = result["per_G"][G]
        r["sequential"]["speedup_est"] = base_seq / r["sequential"]["step_ms_est"]
        r["pipeline"]["speedup_est"] = base_pipe / r["pipeline"]["step_ms_est"]
        r["pipeline"]["ideal_compute_ms"] = result["per_G"][a.gpus[0]]["pipeline"]["compute_ms"] / G * a.gpus[0]
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
