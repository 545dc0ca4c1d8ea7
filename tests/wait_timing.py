"""Where the 256x256 GEMM's k loop waits (diagnostic; needs a -DAC_TILE_TIMING build, see tile_timing.py).

    AC_LIB=$PWD/build_ablate/libac_timing.so python tests/wait_timing.py 48

(the timing library is built with -DAC_EXPERIMENTAL -DAC_TILE_TIMING: policy 108 routes the all-pairs problem
through the 256x256 kernel, which the product uses for the ViT layers only)

Runs the general all-pairs problem once and prints, per workgroup (medians), the shader-clock cycles of
the k loop and the share wave 0 spent waiting at the X barrier (DMA landed) and the Y barrier (fragment
re-reads done).
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "anomaly-clustering_amd", "models"))
import acnative  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    Z = torch.randn(n, 784, 4096, device=dev)
    Zc = Z.clone()
    acnative.set_gemm_policy(108)
    acnative.pair_min_dist_staged(Z, Zc, 0)
    acnative.pair_min_dist_staged(Z, Zc, 0)
    torch.cuda.synchronize()
    fn = acnative.lib().ac_debug_wait_times
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    buf = np.zeros((4096, 2, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    for w in (0, 1):
        t = buf[:, w][buf[:, w, 3] > 0].astype(np.float64)
        nk = t[0, 3]
        med = np.median(t, axis=0)
        print("wave %d: workgroups stamped %d, stages %d" % (4 * w, len(t), nk))
        print("  k loop %.0f cycles = %.0f per stage (MFMA floor 3072)" % (med[0], med[0] / nk))
        print("  X: counter wait %.0f + barrier %.0f per stage;  Y: counter wait %.0f + barrier %.0f per stage"
              % (med[4] / nk, (med[1] - med[4]) / nk, med[5] / nk, (med[2] - med[5]) / nk))
        print("  one batch of 4 global_load_lds: %.0f cycles" % (med[6] / nk))


if __name__ == "__main__":
    main()
