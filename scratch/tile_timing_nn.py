#!/usr/bin/env python3
"""Per-workgroup phase timing of the first pass of the pruned nearest-neighbour search (gemm5::kernel<EpiNNCand, 8, false, 2,
true>: plain fp16 high halves, 256 x 256 tiles) - diagnostic, needs a -DAC_TILE_TIMING build:

    python anomaly-clustering_amd/models/acnative/build.py --variant timing -DAC_TILE_TIMING
    AC_LIB=$PWD/anomaly-clustering_amd/models/acnative/libac_hip_timing.so python scratch/tile_timing_nn.py [N]

N test images against a bank of N images (784 patches x 4096 each).  Medians over the first 8192 workgroups, in us
(s_memrealtime ticks of 10 ns)."""
import ctypes as c
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "anomaly-clustering_amd", "models"))
import acnative  # noqa: E402

P, D = 784, 4096


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 83
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    base = torch.randn((1, P, D), generator=g) * 0.5                   # images share a texture: near neighbours exist
    Zq = (base + torch.randn((n, P, D), generator=g) * 0.3).to(dev)
    Zk = (base + torch.randn((n, P, D), generator=g) * 0.3).to(dev)
    q, k = acnative.pack_images(Zq), acnative.pack_images(Zk)
    del Zq, Zk
    qh, kh = acnative.h16_from_packed(q), acnative.h16_from_packed(k)
    slack = acnative.nn_slack(acnative.max_sq([k], dev), acnative.max_sq([q], dev))
    lib = acnative.lib()
    lib.ac_debug_tile_times_alpha.argtypes = [c.c_void_p, c.c_int]
    acnative.kernel_events = []

    def run():
        st = acnative.NNState(n * P, dev)
        acnative.nn_scan(st, k, kh, q, qh, slack)
        torch.cuda.synchronize()
        _, e0, e1, *_ = acnative.kernel_events[-1]
        return e0.elapsed_time(e1), st

    run()
    lib.ac_debug_tile_times_alpha(None, 1)
    ms, st = run()
    buf = np.zeros((8, 8192, 4), dtype=np.uint64)
    lib.ac_debug_tile_times_alpha(buf.ctypes.data, 0)
    t = buf[0].astype(np.int64)
    t = t[(t[:, 3] > 0) & (t[:, 0] > 0)]
    us = lambda a: (np.median(a) / 100.0, np.quantile(a, 0.1) / 100.0, np.quantile(a, 0.9) / 100.0)
    tiles = ((n * P + 255) // 256) ** 2
    print("first pass %.3f ms, %d tiles = %.1f per CU -> %.1f us per tile slot; candidates per query: mean %.1f max %d"
          % (ms, tiles, tiles / 256.0, ms * 1e3 / (tiles / 256.0), float(st.count.float().mean()), int(st.count.max())))
    print("stamped workgroups: %d   (median / p10 / p90, us)" % len(t))
    print("  prologue (entry -> first stage visible) %.2f / %.2f / %.2f" % us(t[:, 1] - t[:, 0]))
    print("  k loop                                  %.2f / %.2f / %.2f" % us(t[:, 2] - t[:, 1]))
    print("  epilogue (atomics landed)               %.2f / %.2f / %.2f" % us(t[:, 3] - t[:, 2]))
    print("  whole workgroup                         %.2f / %.2f / %.2f" % us(t[:, 3] - t[:, 0]))


if __name__ == "__main__":
    main()
