"""The CPU-oracle leg of tests/test_configs_full_gpu.py::test_config2_whole_alpha_matrix_against_oracle as a stand-alone
program: BASELINE config 2 (bottle-shaped N = 83, seed-0 ViT-B/8, unsupervised tau = 1) through the oracle - 83 ViT forwards
at the reference's batch size 1, 6 806 ordered cdist pairs, fp64 softmax, weighted average, Ward + scores - about a minute
of host time.  conftest.py starts it as a CPU-only child process when the GPU session begins, so that minute runs BESIDE the
GPU tests instead of inside one of them (round 5: the GPU suite was 460 s of the driver's 900 s); the test waits for the
result file and checks that the child's weights are the weights on the device before it believes any of it.

    python tests/_config2_oracle.py <out.npz>
"""
import os
import sys
import time

os.environ.setdefault("AC_BACKBONE_SEED", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "anomaly-clustering_amd", "models")):
    if p not in sys.path:
        sys.path.insert(0, p)


def weight_sums(sd):
    """One number per tensor, in key order: the sum of the fp32 BIT PATTERNS as integers - exact and independent of the
    order of summation (a floating-point sum changes with the thread count of the process that forms it).  Equal weights
    <=> equal lists (the init is seeded and bit-reproducible)."""
    import torch
    return torch.tensor([int(v.detach().float().contiguous().view(torch.int32).to(torch.int64).sum()) for _, v in sorted(sd.items())],
                        dtype=torch.int64)


def compute(sd, images, labels):
    """-> dict of numpy arrays: Z [n, P, D], w [n, P], alpha [n, 1, P] fp32, X [n, D], scores (NMI, ARI, F1), label ids."""
    import numpy as np
    import torch
    from oracle import vit as ovit, aggregate as oagg, alpha as oalpha, cluster as ocluster
    n = images.shape[0]
    Zs = []
    for i in range(n):                                           # the reference's batch size 1 (main.py:213)
        taps = ovit.forward_taps(sd, images[i:i + 1], taps=(10, 11))
        Zs.append(oagg.aggregate([taps[10], taps[11]], 3, 2048, 4096)[0])
    Zr = torch.stack(Zs).reshape(n, 784, 4096)
    w_ref = oalpha.weights_unsupervised(Zr)                      # utils.py:222-227 for every image
    a64 = torch.cat([oalpha.alpha_from_weights(w, 1.0, 1) for w in w_ref], dim=0)     # utils.py:240-257
    a_ref, X_ref = oalpha.weighted_average(a64, Zr)
    rn, ra, rf, rlabel, _ = ocluster.calculate_metrics(X_ref, labels)
    return {"Z": Zr.numpy(), "w": torch.stack(list(w_ref)).numpy() if not isinstance(w_ref, torch.Tensor) else w_ref.numpy(),
            "alpha": a_ref.numpy(), "X": np.asarray(X_ref), "scores": np.array([rn, ra, rf]), "label": np.asarray(rlabel)}


def main():
    import numpy as np
    import torch
    import patchcore.backbones
    from patchcore.datasets import synthetic
    t0 = time.perf_counter()
    torch.set_num_threads(min(32, os.cpu_count() or 1))          # the box's default (one per hardware thread) is 4x slower here
    backbone = patchcore.backbones.load("dino_vitbase8")          # seeded init, on the CPU
    sd = {k: v.detach().cpu() for k, v in backbone.state_dict().items()}
    labels = synthetic.category_labels("bottle")
    images = synthetic.make_images("bottle", labels)
    res = compute(sd, images, labels)
    res["weight_sums"] = weight_sums(sd).numpy()
    res["seconds"] = np.array([time.perf_counter() - t0])
    tmp = sys.argv[1] + ".tmp.npz"
    np.savez(tmp, **res)
    os.replace(tmp, sys.argv[1])                                 # the reader sees a complete file or none


if __name__ == "__main__":
    main()
