#!/usr/bin/env python3
"""Same-box, same-process A/B of builds of libac_hip.so on the pruned exact nearest-neighbour search (supervised
weights): pass 1 (ac_nn_candidates) + pass 2 (ac_nn_refine) of N test images against a bank of N images.

    python scratch/ab_nn.py [--n 83] [--rounds 5] name=path/to/lib.so name2=path2 ...

Interleaved rounds; prints per library the median time of each pass, the candidates per query and whether the final w
equals the first library's bit for bit (it must: the search is exact whatever the candidate lists were)."""
import argparse
import ctypes as c
import statistics

import torch

P, D, CAP = 784, 4096, 64
INF = 0x7f800000


def bind(path):
    lib = c.CDLL(path)
    vp, i64, i = c.c_void_p, c.c_int64, c.c_int
    lib.ac_pack_hl32.argtypes = [vp, i64, i, i64, vp, vp, vp]
    lib.ac_fill_u32.argtypes = [vp, i64, c.c_uint32, vp]
    lib.ac_fill_u64.argtypes = [vp, i64, c.c_uint64, vp]
    lib.ac_max_f32.argtypes = [vp, i64, vp, vp]
    lib.ac_h16_from_hl32.argtypes = [vp, i64, i, vp, vp]
    lib.ac_nn_slack.argtypes = [vp, vp, vp, vp]
    lib.ac_nn_candidates.argtypes = [vp, vp, i64, vp, vp, i64, i, vp, vp, vp, vp, i, i64, vp]
    lib.ac_nn_refine.argtypes = [vp, i64, vp, i64, i, vp, vp, i, i64, vp, vp, vp, vp, vp]
    lib.ac_nn_finalize.argtypes = [vp, i64, vp, vp]
    lib.ac_last_error.restype = c.c_char_p
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=83)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--slack-scale", type=float, default=1.0,
                    help="multiply the candidate window by this (what a tighter proven bound would give: timing only, "
                         "the result is then no longer guaranteed)")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = [(s.split("=", 1)[0], bind(s.split("=", 1)[1])) for s in a.libs]
    st = torch.cuda.current_stream().cuda_stream
    n = a.n
    g = torch.Generator(device="cpu").manual_seed(3)
    base = torch.randn((1, P, D), generator=g) * 0.5                     # shared texture: near neighbours exist
    l0 = libs[0][1]

    def packed(seed_shift):
        Z = (base + torch.randn((n, P, D), generator=g) * 0.3).to(dev)
        hl = torch.empty((n * P, D * 4), dtype=torch.uint8, device=dev)
        sq = torch.empty((n * P,), dtype=torch.float32, device=dev)
        assert l0.ac_pack_hl32(Z.data_ptr(), n * P, D, D, hl.data_ptr(), sq.data_ptr(), st) == 0
        h16 = torch.empty((n * P, D * 2), dtype=torch.uint8, device=dev)
        assert l0.ac_h16_from_hl32(hl.data_ptr(), n * P, D, h16.data_ptr(), st) == 0
        mx = torch.zeros(1, device=dev)
        assert l0.ac_max_f32(sq.data_ptr(), n * P, mx.data_ptr(), st) == 0
        return hl, sq, h16, mx

    khl, ksq, kh, kmx = packed(0)
    qhl, qsq, qh, qmx = packed(1)
    slack = torch.empty(3, device=dev)
    assert l0.ac_nn_slack(kmx.data_ptr(), qmx.data_ptr(), slack.data_ptr(), st) == 0
    slack[:2] *= a.slack_scale
    rows = n * P
    amin = torch.empty(rows, dtype=torch.int32, device=dev)
    best = torch.empty(rows, dtype=torch.float64, device=dev)
    cand = torch.empty((rows, CAP), dtype=torch.int32, device=dev)
    count = torch.empty(rows, dtype=torch.int32, device=dev)
    oflag = torch.zeros(1, dtype=torch.int32, device=dev)
    orows = torch.zeros(rows, dtype=torch.int32, device=dev)
    w = torch.empty(rows, device=dev)
    t1 = {k: [] for k, _ in libs}
    t2 = {k: [] for k, _ in libs}
    info, ref = {}, None
    for r in range(a.rounds + 1):
        for name, lib in libs:
            assert lib.ac_fill_u32(amin.data_ptr(), rows, INF, st) == 0
            assert lib.ac_fill_u64(best.data_ptr(), rows, 0x7ff0000000000000, st) == 0
            assert lib.ac_fill_u32(count.data_ptr(), rows, 0, st) == 0
            oflag.zero_(); orows.zero_()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            rc = lib.ac_nn_candidates(kh.data_ptr(), ksq.data_ptr(), rows, qh.data_ptr(), qsq.data_ptr(), rows, D, slack.data_ptr(),
                                      amin.data_ptr(), cand.data_ptr(), count.data_ptr(), CAP, 0, st)
            assert rc == 0, lib.ac_last_error()
            e[1].record()
            rc = lib.ac_nn_refine(khl.data_ptr(), rows, qhl.data_ptr(), rows, D, cand.data_ptr(), count.data_ptr(), CAP, 0,
                                  best.data_ptr(), oflag.data_ptr(), orows.data_ptr(), slack.data_ptr(), st)
            assert rc == 0, lib.ac_last_error()
            e[2].record()
            assert lib.ac_nn_finalize(best.data_ptr(), rows, w.data_ptr(), st) == 0
            torch.cuda.synchronize()
            if r:
                t1[name].append(e[0].elapsed_time(e[1]))
                t2[name].append(e[1].elapsed_time(e[2]))
            else:
                if ref is None:
                    ref = w.clone()
                info[name] = (float(count.float().mean()), int(count.max()), int(oflag.item()), bool(torch.equal(w, ref)))
    for name, _ in libs:
        m, mx, of, same = info[name]
        print("%-10s pass 1 median %.3f ms (min %.3f max %.3f)  pass 2 %.3f ms  candidates per query %.1f (max %d, overflow %d)  w bit-equal to %s: %s"
              % (name, statistics.median(t1[name]), min(t1[name]), max(t1[name]), statistics.median(t2[name]), m, mx, of, libs[0][0], same))


if __name__ == "__main__":
    main()
