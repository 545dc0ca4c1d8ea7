#!/usr/bin/env python3
"""One-direction device-to-device copy rate between GPU pairs of one node (hipMemcpyPeerAsync under torch's copy_):
the number scratch/predict_scaling.py ASSUMES as --link-gbs (50 GB/s).  Needs two or more GPUs; run it on the first
multi-GPU lease and feed the result back:  python scratch/p2p_probe.py && python scratch/predict_scaling.py --all15 --link-gbs <rate>

Prints one JSON object: GB/s for 0->1 alone, and for 0->1, 0->2, 0->3, 0->4 running at the same time (the four shard
transfers of a ring step batch on 8 GPUs, each over its own xGMI link)."""
import json
import time

import torch


def main():
    n = torch.cuda.device_count()
    if n < 2:
        raise SystemExit("p2p_probe needs two or more GPUs (found %d)" % n)
    nbytes = 256 << 20
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    peers = list(range(1, min(n, 5)))
    dst = {p: torch.empty(nbytes, dtype=torch.uint8, device="cuda:%d" % p) for p in peers}
    streams = {p: torch.cuda.Stream(device="cuda:0") for p in peers}

    def run(which, reps=10):
        for p in which:
            with torch.cuda.stream(streams[p]):
                dst[p].copy_(src, non_blocking=True)
        for d in range(n):
            torch.cuda.synchronize(d)
        t0 = time.perf_counter()
        for _ in range(reps):
            for p in which:
                with torch.cuda.stream(streams[p]):
                    dst[p].copy_(src, non_blocking=True)
        for d in range(n):
            torch.cuda.synchronize(d)
        return reps * nbytes / (time.perf_counter() - t0) / 1e9

    out = {"bytes": nbytes, "gpus": n, "one_link_GBs": run(peers[:1]),
           "per_link_GBs_with_%d_concurrent" % len(peers): run(peers),
           "peer_access": {p: torch.cuda.can_device_access_peer(0, p) for p in peers}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
