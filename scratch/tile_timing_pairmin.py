#!/usr/bin/env python3
"""Per-workgroup phase timing of the symmetric all-pairs launch (diagnostic; needs a -DAC_TILE_TIMING build):

    python anomaly-clustering_amd/models/acnative/build.py --variant timing -DAC_TILE_TIMING -DAC_XCD_WINDOW=32
    python scratch/tile_timing_pairmin.py anomaly-clustering_amd/models/acnative/libac_hip_timing.so [N]

For the first 8192 workgroups (32 per CU): median time from launch to the first stage visible (prologue), the k loop
and the epilogue, in us (s_memrealtime ticks of 10 ns), and the gap between a workgroup's end and its successor's start
on the same CU slot, estimated from the span."""
import ctypes as c
import sys

import numpy as np
import torch

P, D = 784, 4096


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 83
    lib = c.CDLL(path)
    vp, i64, i = c.c_void_p, c.c_int64, c.c_int
    lib.ac_pack_hl32.argtypes = [vp, i64, i, i64, vp, vp, vp]
    lib.ac_fill_u32.argtypes = [vp, i64, c.c_uint32, vp]
    lib.ac_pair_min_sq_packed.argtypes = [vp, vp, i64, vp, vp, i64, i, i, vp, vp, i, i, vp]
    lib.ac_debug_tile_times_alpha.argtypes = [vp, i]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    Z = (torch.randn((n, P, D), generator=g) * 0.6).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    hl = torch.empty((n * P, D * 4), dtype=torch.uint8, device=dev)
    sq = torch.empty((n * P,), dtype=torch.float32, device=dev)
    assert lib.ac_pack_hl32(Z.data_ptr(), n * P, D, D, hl.data_ptr(), sq.data_ptr(), st) == 0
    mb = torch.empty((n, n * P), dtype=torch.int32, device=dev)

    def run():
        assert lib.ac_fill_u32(mb.data_ptr(), mb.numel(), 0x7f800000, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.ac_pair_min_sq_packed(hl.data_ptr(), sq.data_ptr(), n, hl.data_ptr(), sq.data_ptr(), n, P, D, mb.data_ptr(), None, 0, 1, st) == 0
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    run(); run()
    lib.ac_debug_tile_times_alpha(None, 1)
    ms = run()
    buf = np.zeros((8, 8192, 4), dtype=np.uint64)
    lib.ac_debug_tile_times_alpha(buf.ctypes.data, 0)
    t = buf[0].astype(np.int64)
    t = t[t[:, 3] > 0]
    us = lambda a: (np.median(a) / 100.0, np.quantile(a, 0.1) / 100.0, np.quantile(a, 0.9) / 100.0)
    pro, kl, ep = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
    tiles = (n * P + 255) // 256
    tn = (n * P + 127) // 128
    total = tiles * tn - tiles * (tiles - 1)
    print("launch %.3f ms, %d tiles = %.1f per CU -> %.1f us per tile slot" % (ms, total, total / 256.0, ms * 1e3 / (total / 256.0)))
    print("stamped workgroups: %d   (median / p10 / p90, us)" % len(t))
    print("  prologue (launch -> first stage visible) %.2f / %.2f / %.2f" % pro)
    print("  k loop                                   %.2f / %.2f / %.2f" % kl)
    print("  epilogue                                 %.2f / %.2f / %.2f" % ep)
    whole = us(t[:, 3] - t[:, 0])
    print("  whole workgroup                          %.2f / %.2f / %.2f" % whole)
    if len(sys.argv) > 3:                                    # a strided sample over the whole launch (-DAC_TT_STRIDE=8)
        stride = int(sys.argv[3])
        t0 = t[:, 0].min()
        order = np.argsort(t[:, 0])
        ts = t[order]
        nb = 10
        print("  over the launch (stamped every %d-th workgroup), deciles by start time:" % stride)
        for b in range(nb):
            seg = ts[len(ts) * b // nb: len(ts) * (b + 1) // nb]
            print("    start %7.2f..%7.2f ms  kloop %.2f  epilogue %.2f  whole %.2f us   (wg/ms in this decile: %.0f)"
                  % ((seg[0, 0] - t0) / 1e5, (seg[-1, 0] - t0) / 1e5, np.median(seg[:, 2] - seg[:, 1]) / 100.0,
                     np.median(seg[:, 3] - seg[:, 2]) / 100.0, np.median(seg[:, 3] - seg[:, 0]) / 100.0,
                     len(seg) * stride / max(1e-9, (seg[-1, 0] - seg[0, 0]) / 1e5)))
        full = buf[0].astype(np.int64)
        idx = np.nonzero(full[:, 3] > 0)[0]
        xcd = (idx * stride) % 8
        print("  per XCD (block id %% 8) of the sampled workgroups: first start, last end (ms), median whole (us)")
        for x in range(8):
            sel = full[idx[xcd == x]]
            if len(sel):
                print("    xcd slot %d: n %5d  first start %.3f  last end %.3f  whole %.2f  kloop %.2f"
                      % (x, len(sel), (sel[:, 0].min() - t0) / 1e5, (sel[:, 3].max() - t0) / 1e5,
                         np.median(sel[:, 3] - sel[:, 0]) / 100.0, np.median(sel[:, 2] - sel[:, 1]) / 100.0))
        ends = np.sort(t[:, 3]) - t0
        print("  last workgroup ends at %.3f ms; 99 %% of the sampled workgroups had ended by %.3f ms, 95 %% by %.3f ms"
              % (ends[-1] / 1e5, np.quantile(ends, 0.99) / 1e5, np.quantile(ends, 0.95) / 1e5))
        return
    # successive workgroups on one CU slot: sort starts; the r-th round starts when a slot of round r-1 ends
    starts = np.sort(t[:, 0]) - t[:, 0].min()
    ends = np.sort(t[:, 3]) - t[:, 0].min()
    k = min(len(starts) - 256, len(ends))
    gap = (starts[256:256 + k] - ends[:k]) / 100.0          # i-th end frees the slot the (256 + i)-th start takes
    print("  end of a workgroup -> start of the next on the freed CU: median %.2f us (p10 %.2f, p90 %.2f)"
          % (np.median(gap), np.quantile(gap, 0.1), np.quantile(gap, 0.9)))


if __name__ == "__main__":
    main()
