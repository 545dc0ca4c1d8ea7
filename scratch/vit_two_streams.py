#!/usr/bin/env python3
"""Does a ViT forward + aggregation run faster as two half-batches on two HIP streams (the kernel-boundary bubbles and tile
tails of one stream filled by the other's kernels)?  Same images, same results; only time may move.

    python scratch/vit_two_streams.py [83 166 250]"""
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


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [83, 166, 250]
    dev = torch.device("cuda:0")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=patchcore.backbones.load("dino_vitbase8"), layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
    g = torch.Generator().manual_seed(1)
    images = torch.randn((max(sizes), 3, 224, 224), generator=g).to(dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def one(x):
        return core.embed_device(x, batch_size=x.shape[0])

    def two(x):
        h = (x.shape[0] + 1) // 2
        s1.wait_stream(main_s)
        s2.wait_stream(main_s)
        with torch.cuda.stream(s1):
            a = one(x[:h])
        with torch.cuda.stream(s2):
            b = one(x[h:])
        main_s.wait_stream(s1)
        main_s.wait_stream(s2)
        return a, b

    for b in sizes:
        x = images[:b]
        ref = one(x)
        a, c = two(x)
        torch.cuda.synchronize()
        same = bool(torch.equal(torch.cat([a, c]), ref))
        res = {}
        for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(x)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.setdefault(name, []).append(statistics.median(ts))
        print("b = %3d  one stream %s ms   two streams of %d + %d: %s ms   bit-equal: %s"
              % (b, " / ".join("%.3f" % t for t in res["one stream"]), (b + 1) // 2, b // 2,
                 " / ".join("%.3f" % t for t in res["two streams"]), same), flush=True)


if __name__ == "__main__":
    main()
