#!/usr/bin/env python3
"""Same-box, same-process A/B of builds of libac_hip.so on the patch aggregation (ac_patch_aggregate, DINO ViT-B/8 shape:
two [B, 785, 768] token maps, patch size 3, 2 x 2048 -> 4096).

    python scratch/ab_agg.py [--images 83] [--rounds 9] name=path/to/lib.so name2=path2 ...

Prints median / min time per library and whether Z is bit-equal to the first library's."""
import argparse
import ctypes as c
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "anomaly-clustering_amd", "models")]
import torch  # noqa: E402
from acnative import SIGNATURES  # noqa: E402


def bind(path):
    lib = c.CDLL(path)
    for name in ("ac_patch_aggregate_workspace_bytes", "ac_patch_aggregate", "ac_last_error"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = SIGNATURES[name]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=83)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(9)
    B, T, C = a.images, 785, 768
    feats = [(torch.randn((B, T, C), generator=g) * 1.3 + 0.2).to(dev) for _ in range(2)]
    n = 2
    ptrs = (c.c_void_p * n)(*[f.data_ptr() for f in feats])
    kinds, Cs, Hs, Ws = (c.c_int * n)(0, 0), (c.c_int * n)(C, C), (c.c_int * n)(28, 28), (c.c_int * n)(28, 28)
    libs = [(s.split("=", 1)[0], bind(s.split("=", 1)[1])) for s in a.libs]
    Z = torch.empty((B * 784, 4096), dtype=torch.float32, device=dev)
    ref, same, times = None, {}, {k: [] for k, _ in libs}
    for r in range(a.rounds + 1):
        for name, lib in libs:
            nbytes = lib.ac_patch_aggregate_workspace_bytes(n, Cs, Hs, Ws, B, 3, 2048, 4096)
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            Z.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ac_patch_aggregate(ptrs, kinds, Cs, Hs, Ws, n, B, 3, 2048, 4096, 1e-5, Z.data_ptr(), ws.data_ptr(), nbytes, st)
            e1.record()
            assert rc == 0, lib.ac_last_error()
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1))
            else:
                if ref is None:
                    ref = Z.clone()
                same[name] = bool(torch.equal(Z, ref))
    for name, _ in libs:
        t = times[name]
        print("%-10s aggregation of %d images: median %.4f ms  min %.4f  max %.4f   bit-equal to %s: %s"
              % (name, B, statistics.median(t), min(t), max(t), libs[0][0], same[name]))


if __name__ == "__main__":
    main()
