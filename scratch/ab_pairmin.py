#!/usr/bin/env python3
"""Same-box, same-process A/B of builds of libac_hip.so on the all-pairs kernel (and optionally a ViT block).

    python scratch/ab_pairmin.py [--n 83] [--rounds 5] name=path/to/lib.so name2=path2 ...

Every library is loaded with ctypes side by side; the symmetric all-pairs launch (ac_pair_min_sq_packed on the
same packed random Z, N images x 784 patches x 4096) runs `rounds` interleaved rounds per library; prints the
median / min launch time per library and whether the minima are bit-equal to the first library's."""
import argparse
import ctypes as c
import statistics
import sys

import torch

P, D = 784, 4096
INF = 0x7f800000


def bind(path):
    lib = c.CDLL(path)
    vp, i64, i = c.c_void_p, c.c_int64, c.c_int
    lib.ac_pack_hl32.argtypes = [vp, i64, i, i64, vp, vp, vp]
    lib.ac_fill_u32.argtypes = [vp, i64, c.c_uint32, vp]
    lib.ac_pair_min_sq_packed.argtypes = [vp, vp, i64, vp, vp, i64, i, i, vp, vp, i, i, vp]
    lib.ac_last_error.restype = c.c_char_p
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=83)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--no-vendor", action="store_true", help="skip the vendor-library reference GEMMs at the end")
    ap.add_argument("--one-way", type=lambda v: [int(x) for x in v.split(",") if x], default=[],
                    help="also time ONE-WAY blocks (all N key images x the first Q query images) on the first library: what "
                         "re-measuring Q/N of the query patches through the accurate kernel would cost")
    ap.add_argument("--two-way", type=int, default=0,
                    help="instead of the symmetric launch: a TWO-WAY block, keys = the first n images, queries = this many more")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = [(s.split("=", 1)[0], bind(s.split("=", 1)[1])) for s in a.libs]
    g = torch.Generator(device="cpu").manual_seed(3)
    nz = a.n + a.two_way
    Z = torch.empty((nz, P, D), device=dev)
    for s0 in range(0, nz, 64):                          # generated in slices: N = 1200 is 15 GB
        Z[s0:s0 + 64] = (torch.randn((min(64, nz - s0), P, D), generator=g) * 0.6).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    hl = torch.empty((nz * P, D * 4), dtype=torch.uint8, device=dev)
    sq = torch.empty((nz * P,), dtype=torch.float32, device=dev)
    assert libs[0][1].ac_pack_hl32(Z.data_ptr(), nz * P, D, D, hl.data_ptr(), sq.data_ptr(), st) == 0
    del Z
    if a.two_way:
        nq = a.two_way
        mb = torch.empty((a.n, nq * P), dtype=torch.int32, device=dev)
        mbt = torch.empty((nq, a.n * P), dtype=torch.int32, device=dev)
        qhl, qsq = hl[a.n * P:], sq[a.n * P:]
    else:
        mb = torch.empty((a.n, a.n * P), dtype=torch.int32, device=dev)
    times = {name: [] for name, _ in libs}
    ref = None
    same = {}
    for r in range(a.rounds + 1):                      # round 0 = warm-up
        for name, lib in libs:
            assert lib.ac_fill_u32(mb.data_ptr(), mb.numel(), INF, st) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if a.two_way:
                assert lib.ac_fill_u32(mbt.data_ptr(), mbt.numel(), INF, st) == 0
                e0.record()
                rc = lib.ac_pair_min_sq_packed(hl.data_ptr(), sq.data_ptr(), a.n, qhl.data_ptr(), qsq.data_ptr(), nq, P, D,
                                               mb.data_ptr(), mbt.data_ptr(), 0, 1, st)
            else:
                rc = lib.ac_pair_min_sq_packed(hl.data_ptr(), sq.data_ptr(), a.n, hl.data_ptr(), sq.data_ptr(), a.n, P, D,
                                               mb.data_ptr(), None, 0, 1, st)
            e1.record()
            assert rc == 0, lib.ac_last_error()
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1))
            else:
                if ref is None:
                    ref = mb.clone()
                same[name] = bool(torch.equal(mb, ref))
    flop = (2.0 * a.n * a.two_way if a.two_way else a.n * (a.n - 1.0)) * 2 * P * P * D
    for name, _ in libs:
        t = times[name]
        print("%-12s median %.3f ms  min %.3f  max %.3f  (%.0f TFLOP/s algorithmic at the median)  bit-equal to %s: %s"
              % (name, statistics.median(t), min(t), max(t), flop / statistics.median(t) / 1e9, libs[0][0], same[name]))
    for q in a.one_way:
        lib = libs[0][1]
        mq = torch.empty((a.n, q * P), dtype=torch.int32, device=dev)
        ts = []
        for r in range(a.rounds + 1):
            assert lib.ac_fill_u32(mq.data_ptr(), mq.numel(), INF, st) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ac_pair_min_sq_packed(hl.data_ptr(), sq.data_ptr(), a.n, hl.data_ptr(), sq.data_ptr(), q, P, D,
                                           mq.data_ptr(), None, 0, 1, st)
            e1.record()
            assert rc == 0, lib.ac_last_error()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        print("one-way block, %d key images x %d query images (%.1f %% of the query patches): median %.3f ms  min %.3f"
              % (a.n, q, 100.0 * q / a.n, statistics.median(ts), min(ts)))
    if a.no_vendor:
        return 0
    # vendor reference on the same box: fp16 GEMM through torch (hipBLASLt / rocBLAS), random operands
    # ... and at the shapes of the ViT linear layers at 83 images (qkv, proj, fc1, fc2)
    for (m, n, k) in [(8192, 8192, 8192), (32768, 32768, 4096), (65155, 2304, 768), (65155, 768, 768), (65155, 3072, 768),
                      (65155, 768, 3072)]:
        A = torch.randn((m, k), device=dev, dtype=torch.float16)
        B = torch.randn((n, k), device=dev, dtype=torch.float16)
        C = A @ B.T
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            C = A @ B.T
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("torch fp16 matmul %dx%dx%d (vendor library, random operands): median %.3f ms = %.0f TFLOP/s"
              % (m, n, k, statistics.median(ts), 2.0 * m * n * k / statistics.median(ts) / 1e9))
        del A, B, C
    # vendor attention (torch SDPA: the library's fused fp16 kernel) at the ViT-B/8 shape of 83 images
    try:
        import torch.nn.functional as F
        q, k, v = (torch.randn((83, 12, 785, 64), device=dev, dtype=torch.float16) for _ in range(3))
        o = F.scaled_dot_product_attention(q, k, v)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o = F.scaled_dot_product_attention(q, k, v)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        fl = 83 * 12 * 4.0 * 785 * 785 * 64
        print("torch SDPA fp16 [83,12,785,64] (single pass): median %.3f ms = %.0f TFLOP/s" % (statistics.median(ts), fl / statistics.median(ts) / 1e9))
    except Exception as e:          # noqa: BLE001
        print("torch SDPA not available:", e)


if __name__ == "__main__":
    sys.exit(main())
