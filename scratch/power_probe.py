#!/usr/bin/env python3
"""What the chip draws and clocks while the all-pairs kernel runs back to back - on real operands and on zeroed ones.
Polls `rocm-smi` (socket power, sclk, power cap) from the host while a thread keeps the launch queue full.

    python scratch/power_probe.py [--n 83] [--seconds 6]"""
import argparse
import ctypes as c
import json
import os
import statistics
import subprocess
import threading
import time

import torch

P, D = 784, 4096
HERE = os.path.dirname(os.path.abspath(__file__))


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True,
                             timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(k for k in d if k.startswith("card"))[0]]
        return card
    except Exception as e:                                      # noqa: BLE001 - diagnostic script
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=83)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--more", action="store_true", help="also the ViT forward and the vendor library's fp16 GEMM")
    ap.add_argument("--lib", default=os.path.join(HERE, "..", "anomaly-clustering_amd", "models", "acnative", "libac_hip.so"),
                    help="another build of the library (e.g. an -DAC_ABLATE variant: where the joules go)")
    ap.add_argument("--only-real", action="store_true", help="only the all-pairs kernel on real operands")
    a = ap.parse_args()
    lib = c.CDLL(a.lib)
    vp, i64, i = c.c_void_p, c.c_int64, c.c_int
    lib.ac_pack_hl32.argtypes = [vp, i64, i, i64, vp, vp, vp]
    lib.ac_fill_u32.argtypes = [vp, i64, c.c_uint32, vp]
    lib.ac_pair_min_sq_packed.argtypes = [vp, vp, i64, vp, vp, i64, i, i, vp, vp, i, i, vp]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    n = a.n
    g = torch.Generator(device="cpu").manual_seed(3)
    Z = (torch.randn((n, P, D), generator=g) * 0.6).to(dev)
    hl = torch.empty((n * P, D * 4), dtype=torch.uint8, device=dev)
    sq = torch.empty((n * P,), dtype=torch.float32, device=dev)
    assert lib.ac_pack_hl32(Z.data_ptr(), n * P, D, D, hl.data_ptr(), sq.data_ptr(), st) == 0
    del Z
    zeros = torch.zeros_like(hl)
    mb = torch.empty((n, n * P), dtype=torch.int32, device=dev)
    print("idle:", json.dumps(smi()))

    def pair(h):
        return lambda: lib.ac_pair_min_sq_packed(h.data_ptr(), sq.data_ptr(), n, h.data_ptr(), sq.data_ptr(), n, P, D, mb.data_ptr(),
                                                 None, 0, 1, st)
    jobs = [("all-pairs kernel, real operands", pair(hl)), ("all-pairs kernel, zeroed operands", pair(zeros))]
    if a.only_real:
        jobs = jobs[:1]
    if a.more:
        import sys
        sys.path.insert(0, os.path.join(HERE, "..", "anomaly-clustering_amd", "models"))
        os.environ.setdefault("AC_BACKBONE_SEED", "0")
        import patchcore.backbones
        import patchcore.common as common
        import patchcore.sampler
        from patchcore.patchcore import AnomalyClusteringCore
        core = AnomalyClusteringCore(dev)
        core.load(backbone=patchcore.backbones.load("dino_vitbase8"), layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
                  input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
                  featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5, nn_method=common.FaissNN(True, 4))
        imgs = torch.randn((83, 3, 224, 224), generator=g).to(dev)
        jobs.append(("ViT-B/8 forward + aggregation, 83 images", lambda: core.embed_device(imgs, batch_size=83)))
        A = torch.randn((8192, 8192), device=dev, dtype=torch.float16)
        B = torch.randn((8192, 8192), device=dev, dtype=torch.float16)
        jobs.append(("vendor fp16 GEMM 8192^3 (torch.matmul), random operands", lambda: torch.matmul(A, B.T)))
        Az, Bz = torch.zeros_like(A), torch.zeros_like(B)
        jobs.append(("vendor fp16 GEMM 8192^3, zeroed operands", lambda: torch.matmul(Az, Bz.T)))
    for name, fn in jobs:
        stop = threading.Event()
        times = []

        def work():
            while not stop.is_set():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / 4)
        t = threading.Thread(target=work)
        t.start()
        time.sleep(1.0)
        samples = []
        t_end = time.time() + a.seconds
        while time.time() < t_end:
            samples.append(smi())
            time.sleep(0.2)
        stop.set()
        t.join()
        print("%s: %.3f ms per call (median of %d); %d samples" % (name, statistics.median(times), len(times), len(samples)))
        for k in ("Current Socket Graphics Package Power (W)", "Max Graphics Package Power (W)", "sclk clock speed:"):
            vals = [s.get(k) for s in samples if k in s]
            print("   %-50s %s" % (k, " ".join(str(v) for v in vals[:: max(1, len(vals) // 6)][:6])))


if __name__ == "__main__":
    main()
