#!/usr/bin/env python3
"""Same process, same library, interleaved: a ViT-B/8 forward (12 blocks) with LayerNorm inside the residual GEMMs
(EpiResidual::tail of the patched csrc/vit.hip) against the separate LayerNorm launches (ac_set_ln_fusion(0)).
Outputs compared bit for bit every round.  NEEDS scratch/exp_layernorm_in_residual_gemm.patch applied (the shipped library has
no ac_set_ln_fusion: the experiment was bit-equal and slower - profiles/r05_vit_layernorm_in_residual_gemm_ab.txt).

    python scratch/ab_ln_fusion.py [--images 83] [--rounds 7] [--reps 5]
"""
import argparse
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "anomaly-clustering_amd", "models")]
import torch  # noqa: E402
import acnative  # noqa: E402
import patchcore.vision_transformer as vits  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, nargs="+", default=[83])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", type=int, default=None, help="0 / 1: run only that mode (per-kernel times under rocprofv3)")
    a = ap.parse_args()
    if not hasattr(acnative, "set_ln_fusion"):
        sys.exit("ab_ln_fusion: this library has no ac_set_ln_fusion - apply scratch/exp_layernorm_in_residual_gemm.patch and rebuild "
                 "(the experiment did not ship: profiles/r05_vit_layernorm_in_residual_gemm_ab.txt)")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = vits.vit_base(patch_size=8, num_classes=0).eval().to(dev)
    for n in a.images:
        x = torch.randn(n, 785, 768, device=dev)
        times = {0: [], 1: []}
        outs = {}
        with torch.no_grad():
            for r in range(a.rounds + 1):
                for fused in ((0, 1) if a.only is None else (a.only,)):
                    acnative.set_ln_fusion(fused)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.reps):
                        t = x
                        for blk in m.blocks:
                            t = blk(t)
                    e1.record()
                    torch.cuda.synchronize()
                    outs[fused] = t
                    if r:
                        times[fused].append(e0.elapsed_time(e1) / a.reps)
                if a.only is None:
                    assert torch.equal(outs[0], outs[1]), "fused and separate LayerNorm differ"
        if a.only is not None:
            print("%3d images: 12 blocks  ln_fusion=%d  %.3f ms" % (n, a.only, statistics.median(times[a.only])))
            continue
        t0, t1 = statistics.median(times[0]), statistics.median(times[1])
        print("%3d images: 12 blocks  separate LayerNorm %.3f ms (min %.3f)   inside the GEMMs %.3f ms (min %.3f)   %+.2f %%   bit-equal"
              % (n, t0, min(times[0]), t1, min(times[1]), 100.0 * (t1 / t0 - 1.0)))
    acnative.set_ln_fusion(1)


if __name__ == "__main__":
    main()
