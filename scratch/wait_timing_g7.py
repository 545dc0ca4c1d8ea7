#!/usr/bin/env python3
"""Where the cycles of the 256x192 all-pairs kernel go (a -DAC_G7_TIMING build of the library; gemm_hl7.h).

    python anomaly-clustering_amd/models/acnative/build.py --variant g7timing -DAC_G7_TIMING
    python scratch/wait_timing_g7.py anomaly-clustering_amd/models/acnative/libac_hip_g7timing.so [--n 83]

Runs the symmetric all-pairs launch of scratch/ab_pairmin.py a few times back to back (the chip at its working clock),
then reads the per-workgroup sums of waves 0 (DMA-issuing, older) and 4 (its SIMD partner)."""
import argparse
import ctypes as c
import statistics

import torch

P, D, INF = 784, 4096, 0x7f800000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--n", type=int, default=83)
    a = ap.parse_args()
    lib = c.CDLL(a.lib)
    vp, i64, i = c.c_void_p, c.c_int64, c.c_int
    lib.ac_pack_hl32.argtypes = [vp, i64, i, i64, vp, vp, vp]
    lib.ac_fill_u32.argtypes = [vp, i64, c.c_uint32, vp]
    lib.ac_pair_min_sq_packed.argtypes = [vp, vp, i64, vp, vp, i64, i, i, vp, vp, i, i, vp]
    lib.ac_debug_wait_times7.argtypes = [vp]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    Z = (torch.randn((a.n, P, D), generator=g) * 0.6).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    hl = torch.empty((a.n * P, D * 4), dtype=torch.uint8, device=dev)
    sq = torch.empty((a.n * P,), dtype=torch.float32, device=dev)
    assert lib.ac_pack_hl32(Z.data_ptr(), a.n * P, D, D, hl.data_ptr(), sq.data_ptr(), st) == 0
    del Z
    mb = torch.empty((a.n, a.n * P), dtype=torch.int32, device=dev)
    for r in range(4):
        assert lib.ac_fill_u32(mb.data_ptr(), mb.numel(), INF, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.ac_pair_min_sq_packed(hl.data_ptr(), sq.data_ptr(), a.n, hl.data_ptr(), sq.data_ptr(), a.n, P, D,
                                         mb.data_ptr(), None, 0, 1, st) == 0
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (c.c_uint64 * (1024 * 2 * 8))()
    assert lib.ac_debug_wait_times7(buf) == 0
    rows = [[buf[(b * 2 + w) * 8 + k] for k in range(8)] for b in range(1024) for w in range(2)]
    print("launch %.3f ms (timing build)" % ms)
    for w, name in ((0, "wave 0 (issues the DMA)"), (1, "wave 4 (its SIMD partner)")):
        rs = [r for k, r in enumerate(rows) if k % 2 == w and r[0] > 0]
        if not rs:
            continue
        tiles = sum(r[0] for r in rs)
        nk = rs[0][7] & 0xffff
        e1 = statistics.median((r[7] >> 16) & 0xffffff for r in rs)
        e2 = statistics.median(r[7] >> 40 for r in rs)
        med = lambda k: statistics.median(r[k] / r[0] for r in rs)
        clock = statistics.median(r[6] for r in rs) / (ms * 1e3)        # cycles per microsecond = MHz
        print("%s: %d workgroups, %d tiles, %d stages per tile, ~%.0f MHz" % (name, len(rs), tiles, nk, clock))
        print("  per tile (median over workgroups, shader cycles): prologue %.0f   k loop %.0f (%.0f per stage)   "
              "of it waiting for own DMA at X %.0f (%.1f per stage), for the other waves %.0f (%.1f per stage)   "
              "epilogue + next tile id %.0f (next id, tile arithmetic, early DMA issue %.0f; adding the two accumulators %.0f)"
              % (med(1), med(2), med(2) / nk, med(3), med(3) / nk, med(4), med(4) / nk, med(5), e1, e2))
        print("  MFMA floor per stage: 72 MFMAs x 16 cycles x 2 waves per SIMD = 2304 cycles")


if __name__ == "__main__":
    main()
