"""Ad-hoc GPU timing helper (not a test): python tests/perf_probe.py <what> [args].
   alpha N        - time the all-pairs kernel on N synthetic images, symmetric and general path
   vit B          - time one ViT block at batch B
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "anomaly-clustering_amd", "models")]
import torch  # noqa: E402


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    import acnative
    what = sys.argv[1]
    dev = torch.device("cuda:0")
    if os.environ.get("AC_POLICY"):                  # experimental builds (AC_LIB=...): pin a GEMM generation
        acnative.set_gemm_policy(int(os.environ["AC_POLICY"]))
    if what == "alpha":
        N = int(sys.argv[2])
        g = torch.Generator(device="cpu").manual_seed(0)
        Z = torch.randn(N, 784, 4096, generator=g).to(dev)
        Zc = Z.clone()
        acnative.kernel_events = []
        acnative.pair_min_dist_staged(Z, Z, 0)
        acnative.pair_min_dist_staged(Z, Z, 0)
        acnative.pair_min_dist_staged(Z, Zc, 0)
        acnative.pair_min_dist_staged(Z, Zc, 0)
        torch.cuda.synchronize()
        ms = [e0.elapsed_time(e1) for (_, e0, e1, *_r) in acnative.kernel_events]
        fl = N * N * 2.0 * 784 * 784 * 4096
        print("N=%d symmetric %.2f ms (%.0f TF algorithmic)  general %.2f ms (%.0f TF)"
              % (N, ms[1], fl / ms[1] / 1e9, ms[3], fl / ms[3] / 1e9))
    elif what == "agg":
        B = int(sys.argv[2])
        f10 = torch.randn(B, 785, 768, device=dev)
        f11 = torch.randn(B, 785, 768, device=dev)
        t = timeit(lambda: acnative.patch_aggregate([f10, f11], 3, 2048, 4096), 5)
        print("B=%d aggregate %.3f ms  (%.1f us/img, %.2f TB/s algorithmic at 17.7 MB/img)"
              % (B, t * 1e3, t * 1e6 / B, B * 17.7e6 / t / 1e12))
    elif what == "vit":
        B = int(sys.argv[2])
        import patchcore.vision_transformer as vits
        torch.manual_seed(0)
        m = vits.vit_base(patch_size=8, num_classes=0, depth=1).eval().to(dev)
        x = torch.randn(B, 785, 768, device=dev)
        t = timeit(lambda: m.blocks[0](x), 5)
        print("B=%d one block %.3f ms  (%.1f us/img, %.0f TF algorithmic)" % (B, t * 1e3, t * 1e6 / B, B * 13.0e9 / t / 1e12))


if __name__ == "__main__":
    main()
