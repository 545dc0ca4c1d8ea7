"""GPU: the two linear-layer GEMM kernels of the product must agree BIT FOR BIT.

csrc/gemm_hl2.h (256x128 tiles) and csrc/gemm_hl5.h (256x256 tiles) accumulate the same three split-fp16
products per 32-deep k step in the same order into one fp32 accumulator, so the tiling is invisible in
the result.  By default the library picks by problem size (small problems take the 256x128 kernel), so the
small parity cases of test_vit_gpu.py would never reach the 256x256 kernel; here ac_set_gemm_policy() pins
each kernel in turn on the same inputs and the outputs are compared exactly:
  * a ViT-B/8 block at batch 3 (row tail: 2355 tokens; K = 768 and 3072),
  * a narrow transformer (embed 64, 1 head, hidden 128: K = 64 -> two k stages, K = 128 -> four) at
    batch 5, and its patch embedding (K = 192).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_all(dev):
    import patchcore.vision_transformer as vits
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
    return out


def test_generations_agree_bitwise(dev):
    import acnative
    try:
        acnative.set_gemm_policy(acnative.GEMM_256x128)
        ref = _run_all(dev)
        acnative.set_gemm_policy(acnative.GEMM_256x256)
        big = _run_all(dev)
        acnative.set_gemm_policy(acnative.GEMM_AUTO)
        auto = _run_all(dev)
    finally:
        acnative.set_gemm_policy(acnative.GEMM_AUTO)
    for k in ref:
        assert np.isfinite(ref[k]).all(), k
        np.testing.assert_array_equal(big[k], ref[k], err_msg="256x256 vs 256x128: %s" % k)
        np.testing.assert_array_equal(auto[k], ref[k], err_msg="auto vs 256x128: %s" % k)


def test_unknown_policy_is_rejected():
    import acnative
    with pytest.raises(RuntimeError, match="unknown policy"):
        acnative.set_gemm_policy(77)
