#!/usr/bin/env python3
"""Does the all-pairs kernel run faster when its operands toggle fewer bits?  (The chip is power-limited under this
kernel: DESIGN.md section 4.)  Same library, same launch, interleaved rounds; only the BYTES of the packed operands differ:

    full        hl32 as packed (hi = fp16(x), lo = fp16(x - hi): 10 random mantissa bits in each plane)
    lo-m<k>     the low k mantissa bits of every lo half zeroed (k = 2, 4, 6, 8)
    lo-zero     the lo plane all zero (single-fp16 operands, three passes still issued)
    zero        every operand byte zero

    python scratch/exp_operand_bits.py [--n 83] [--rounds 5] [--lib path]

Prints the median launch time per variant.  This is a measurement of the hardware, not a shipped mode: the variants
other than `full` compute different numbers."""
import argparse
import ctypes as c
import os
import statistics

import torch

P, D = 784, 4096
INF = 0x7f800000
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=83)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--lib", default=os.path.join(HERE, "..", "anomaly-clustering_amd", "models", "acnative", "libac_hip.so"))
    a = ap.parse_args()
    lib = c.CDLL(a.lib)
    vp, i64, i = c.c_void_p, c.c_int64, c.c_int
    lib.ac_pack_hl32.argtypes = [vp, i64, i, i64, vp, vp, vp]
    lib.ac_fill_u32.argtypes = [vp, i64, c.c_uint32, vp]
    lib.ac_pair_min_sq_packed.argtypes = [vp, vp, i64, vp, vp, i64, i, i, vp, vp, i, i, vp]
    lib.ac_last_error.restype = c.c_char_p
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(3)
    n = a.n
    Z = torch.empty((n, P, D), device=dev)
    for s0 in range(0, n, 64):
        Z[s0:s0 + 64] = (torch.randn((min(64, n - s0), P, D), generator=g) * 0.6).to(dev)
    hl = torch.empty((n * P, D * 4), dtype=torch.uint8, device=dev)
    sq = torch.empty((n * P,), dtype=torch.float32, device=dev)
    assert lib.ac_pack_hl32(Z.data_ptr(), n * P, D, D, hl.data_ptr(), sq.data_ptr(), st) == 0
    del Z

    def variant(kind):
        v = hl.clone()
        halves = v.view(torch.int16).view(n * P, D // 32, 64)          # per 128-byte group: 32 hi halves, 32 lo halves
        if kind.startswith("lo-m"):
            k = int(kind[4:])
            halves[:, :, 32:] &= ~((1 << k) - 1)
        elif kind == "lo-zero":
            halves[:, :, 32:] = 0
        elif kind == "zero":
            v.zero_()
        return v

    kinds = ["full", "lo-m2", "lo-m4", "lo-m5", "lo-m6", "lo-m8", "lo-zero", "zero"]
    packed = {k: variant(k) for k in kinds}
    mb = torch.empty((n, n * P), dtype=torch.int32, device=dev)
    times = {k: [] for k in kinds}
    for r in range(a.rounds + 1):
        for k in kinds:
            h = packed[k]
            assert lib.ac_fill_u32(mb.data_ptr(), mb.numel(), INF, st) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ac_pair_min_sq_packed(h.data_ptr(), sq.data_ptr(), n, h.data_ptr(), sq.data_ptr(), n, P, D,
                                           mb.data_ptr(), None, 0, 1, st)
            e1.record()
            assert rc == 0, lib.ac_last_error()
            torch.cuda.synchronize()
            if r:
                times[k].append(e0.elapsed_time(e1))
    base = statistics.median(times["full"])
    for k in kinds:
        t = statistics.median(times[k])
        print("%-8s median %.3f ms  min %.3f  max %.3f   %+.1f %% against full" % (k, t, min(times[k]), max(times[k]), 100.0 * (t / base - 1.0)))


if __name__ == "__main__":
    main()
