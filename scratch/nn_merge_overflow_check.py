#!/usr/bin/env python3
"""Round 6: does scanning the remote bank shards of a G-rank supervised run as ONE key block (pipeline._SupJob.run_later) fill a query's
candidate list (acnative.NN_CAP slots per scan) more than scanning them shard by shard does?  Config-4-shaped data (a category of N test
images against a bank of N normal images, real ViT embeddings of synthetic images): rank 0's queries against the bank in G shards, once
shard by shard (own shard first, as the ring delivers them), once own shard + ONE block of the rest.  Prints the candidate counts and the
overflow flag of both; the weights must agree to the bit.

    python scratch/nn_merge_overflow_check.py [category] [G]"""
import os
import sys

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
from patchcore.pipeline import HipOps, shard_bounds  # noqa: E402


def main():
    cat = sys.argv[1] if len(sys.argv) > 1 else "bottle"
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=patchcore.backbones.load("dino_vitbase8"), layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    ops = HipOps(core, batch_size=128)
    labels = synthetic.category_labels(cat)
    n = len(labels)
    test = synthetic.make_images(cat, labels).to(dev)
    bank = synthetic.make_images(cat, ["good"] * n, split=synthetic.DatasetSplit.TRAIN).to(dev)
    qb, bb = shard_bounds(n, G), shard_bounds(n, G)
    pk_q = ops.pack(ops.embed(test[qb[0][0]:qb[0][1]]))                     # rank 0's queries
    pk_b = ops.pack(ops.embed(bank))
    slack = ops.nn_slack(ops.max_sq([pk_b], dev), ops.max_sq([pk_q], dev))
    q16 = ops.h16(pk_q)
    P = pk_q.P
    out = {}
    for name in ("shard by shard", "own shard + one block"):
        st = ops.nn_state(pk_q.n * P, dev)
        blocks = [pk_b.slice(*bb[k]) for k in range(G)] if name == "shard by shard" else [pk_b.slice(*bb[0]), pk_b.slice(bb[1][0], n)]
        counts = []
        for blk in blocks:
            if blk.n:
                ops.nn_scan(st, blk, ops.h16(blk), pk_q, q16, slack)
                counts.append(st.count.clone())
        torch.cuda.synchronize()
        c = torch.stack(counts).float()
        out[name] = ops.nn_finalize(st, P)
        print("%-22s scans %d  candidates per query and scan: mean %.1f  max %d  (cap %d)  overflow flag %d  overflowed queries %d"
              % (name, len(counts), c.mean().item(), int(c.max().item()), st.cap, int(st.overflow.item()), int((st.overflow_rows != 0).sum().item())))
    print("weights bit-equal:", bool(torch.equal(out["shard by shard"], out["own shard + one block"])))


if __name__ == "__main__":
    main()
