#!/usr/bin/env python3
"""ViT-B/8 forward + aggregation: time per image as a function of the images per launch (tile-round quantisation:
rows = 785 b, 256-row tiles, 256 CUs).

    python scratch/vit_batch_sweep.py 83 84 96 128 166 167 ...

Prints ms per batch and us per image (median of 5) and the predicted tile rounds of the four linear layers."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "anomaly-clustering_amd", "models"))
os.environ.setdefault("AC_BACKBONE_SEED", "0")
import patchcore.backbones  # noqa: E402
import patchcore.common as common  # noqa: E402
import patchcore.sampler  # noqa: E402
from patchcore.patchcore import AnomalyClusteringCore  # noqa: E402


def rounds(b):
    tm = math.ceil(b * 785 / 256)
    return [tm * tn / 256.0 for tn in (9, 3, 12, 3)]


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [83, 84, 96, 128, 166, 167]
    dev = torch.device("cuda:0")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=patchcore.backbones.load("dino_vitbase8"), layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    g = torch.Generator().manual_seed(1)
    images = torch.randn((max(sizes), 3, 224, 224), generator=g).to(dev)
    for b in sizes:
        x = images[:b]
        core.embed_device(x, batch_size=b)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            core.embed_device(x, batch_size=b)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = statistics.median(ts)
        print("b = %3d  %.3f ms  %.1f us per image   tile rounds qkv/proj/fc1/fc2: %s"
              % (b, t, 1e3 * t / b, " ".join("%.2f" % r for r in rounds(b))), flush=True)


if __name__ == "__main__":
    main()
