#!/usr/bin/env python3
"""Same-box, same-process A/B of builds of libac_hip.so on ONE ViT-B/8 block (ac_vit_block) at a batch of images.

    python scratch/ab_vit.py [--images 83] [--rounds 7] name=path/to/lib.so name2=path2[@policy] ...

`@policy`: ac_set_gemm_policy(policy) on that library before its runs (1 / 2 pin the 256x128 / 256x256 kernel; 101 ... 109
select a superseded GEMM generation of a `build.py --experimental` library, e.g. 101 = round 1's 128x128 tiles on 4 waves with
64 KB of LDS: TWO co-resident workgroups per CU).

Random block weights (packed by each library itself), random residual stream; interleaved rounds; prints median /
min time of the block per library and whether the outputs are bit-equal to the first library's."""
import argparse
import ctypes as c
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "anomaly-clustering_amd", "models")]
import torch  # noqa: E402
from acnative import BlockWeights, SIGNATURES  # noqa: E402  (ctypes struct + signature table only; no library is loaded)

T, C, H, HID = 785, 768, 12, 3072


def bind(path):
    lib = c.CDLL(path)
    for name in ("ac_pack_hl32", "ac_vit_workspace_bytes", "ac_vit_block", "ac_last_error"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = SIGNATURES[name]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=83)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    rnd = lambda *shape, s=0.02: (torch.randn(shape, generator=g) * s).to(dev)
    mats = {"qkv_w": rnd(3 * C, C), "proj_w": rnd(C, C), "fc1_w": rnd(HID, C), "fc2_w": rnd(C, HID)}
    vecs = {"ln1_w": 1 + rnd(C, s=0.1), "ln1_b": rnd(C, s=0.1), "qkv_b": rnd(3 * C, s=0.1), "proj_b": rnd(C, s=0.1),
            "ln2_w": 1 + rnd(C, s=0.1), "ln2_b": rnd(C, s=0.1), "fc1_b": rnd(HID, s=0.1), "fc2_b": rnd(C, s=0.1)}
    x = rnd(a.images, T, C, s=1.0)
    libs = []
    for spec in a.libs:
        name, path = spec.split("=", 1)
        policy = None
        if "@" in path:
            path, policy = path.rsplit("@", 1)
        lib = bind(path)
        if policy is not None:
            assert lib.ac_set_gemm_policy(int(policy)) == 0, lib.ac_last_error()
        keep, w = {}, BlockWeights()
        for k, m in mats.items():
            out = torch.empty((m.shape[0], m.shape[1] * 4), dtype=torch.uint8, device=dev)
            assert lib.ac_pack_hl32(m.data_ptr(), m.shape[0], m.shape[1], m.shape[1], out.data_ptr(), None, st) == 0
            keep[k] = out
            setattr(w, k, out.data_ptr())
        for k, v in vecs.items():
            setattr(w, k, v.data_ptr())
        n = lib.ac_vit_workspace_bytes(a.images, T, C, HID, H, 32)
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        libs.append((name, lib, w, keep, ws, n))
    times = {n: [] for n, *_ in libs}
    ref, same = None, {}
    for r in range(a.rounds + 1):
        for name, lib, w, keep, ws, n in libs:
            y = torch.empty_like(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ac_vit_block(c.byref(w), x.data_ptr(), y.data_ptr(), a.images, T, C, H, HID, 1e-6, ws.data_ptr(), n, st)
            e1.record()
            assert rc == 0, lib.ac_last_error()
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1))
            else:
                if ref is None:
                    ref = y.clone()
                same[name] = bool(torch.equal(y, ref))
    for name, *_ in libs:
        t = times[name]
        print("%-12s ViT block at %d images: median %.4f ms  min %.4f  max %.4f   bit-equal to %s: %s"
              % (name, a.images, statistics.median(t), min(t), max(t), libs[0][0], same[name]))


if __name__ == "__main__":
    sys.exit(main())
