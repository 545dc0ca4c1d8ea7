"""GPU: the GEMM generations must agree BIT FOR BIT.

Every generation (csrc/gemm_hl2.h 256x128, gemm_hl5.h 256x256, ...) accumulates the same three
split-fp16 products per 32-deep k step in the same order into one fp32 accumulator, so the tiling is
invisible in the result.  The library picks the kernel from AC_GEMM once per process and, by default,
by problem size (small problems fall back to the 256x128 kernel) - so the small parity cases of
test_vit_gpu.py would never reach the 256x256 kernel.  Here each generation runs the same blocks in
its own process and the outputs are compared exactly:
  * a ViT-B/8 block at batch 3 (row tail: 2355 tokens; K = 768 and 3072),
  * a narrow transformer (embed 64, 1 head, hidden 128: K = 64 -> two k stages, K = 128 -> four) at
    batch 5, and its patch embedding (K = 192),
  * the all-pairs distance (symmetric and general) at D = 1024.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = os.path.join(HERE, "..", "anomaly-clustering_amd", "models")

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(models)r)
import patchcore.vision_transformer as vits
dev = torch.device("cuda:0")
out = {}
torch.manual_seed(1)
m = vits.vit_base(patch_size=8, num_classes=0, depth=1).eval().to(dev)
x = torch.randn(3, 785, 768, device=dev)
with torch.no_grad():
    out["base"] = m.blocks[0](x).cpu().numpy()
torch.manual_seed(2)
n = vits.VisionTransformer(img_size=[64], patch_size=8, in_chans=3, num_classes=0, embed_dim=64, depth=2,
                           num_heads=1, mlp_ratio=2, qkv_bias=True).eval().to(dev)
img = torch.randn(5, 3, 64, 64, device=dev)
with torch.no_grad():
    t = n.prepare_tokens(img)
    out["narrow_tokens"] = t.cpu().numpy()
    for blk in n.blocks:
        t = blk(t)
    out["narrow"] = t.cpu().numpy()
import acnative
torch.manual_seed(3)
Z = torch.randn(5, 49, 1024, device=dev) + torch.randn(1, 1, 1024, device=dev) * 3.0
out["alpha_w_sym"] = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS).cpu().numpy()
out["alpha_w_gen"] = acnative.pair_min_dist(Z[:3], Z[1:].contiguous(), acnative.REDUCE_MIN_ALL).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def _run(gen, path):
    env = dict(os.environ, AC_GEMM=str(gen))
    r = subprocess.run([sys.executable, "-c", CHILD % {"models": os.path.abspath(MODELS)}, path], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "AC_GEMM=%s failed:\n%s\n%s" % (gen, r.stdout[-2000:], r.stderr[-2000:])
    return np.load(path)


def test_generations_agree_bitwise(tmp_path):
    ref = _run(3, str(tmp_path / "g3.npz"))            # 256x128 ring kernel everywhere
    for gen in (10, 8):                                # 256x256 everywhere / the default size-based choice
        got = _run(gen, str(tmp_path / ("g%d.npz" % gen)))
        for k in ref.files:
            if k.startswith("alpha") and gen == 8:     # AC_GEMM=8 runs the all-pairs kernel without the
                continue                               # second accumulator (timing studies only)
            assert np.isfinite(got[k]).all(), (gen, k)
            np.testing.assert_array_equal(got[k], ref[k], err_msg="AC_GEMM=%d vs 3: %s" % (gen, k))


def test_all_pairs_two_level_variants_agree_bitwise(tmp_path):
    """The all-pairs kernel with two-level accumulation: 256x128 ring kernel (default) vs the gemm_hl5
    stream at 256x128 (AC_GEMM=9) - same flush schedule, same sums (D = 1024: two flushes)."""
    ref = _run(3, str(tmp_path / "a3.npz"))
    for gen in (9, 11):                                # 11: the barrier-free (LDS counter) variant, gemm_hl6.h
        got = _run(gen, str(tmp_path / ("a%d.npz" % gen)))
        for k in ("alpha_w_sym", "alpha_w_gen"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg="AC_GEMM=%d: %s" % (gen, k))
