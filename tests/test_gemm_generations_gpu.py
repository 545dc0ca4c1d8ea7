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


def test_all_pairs_tiles_agree_bitwise(dev):
    """Round 5: the all-pairs distance kernel runs on 256x192 tiles (csrc/gemm_hl7.h); ac_set_gemm_policy(256x128) pins the
    round-4 kernel (csrc/gemm_hl2.h).  Per element both accumulate the same chain (k stages in order, passes hi*lo, lo*hi,
    hi*hi, the MFMA accumulator flushed into the second one every 16 stages), every pair is formed with the lower row as
    the A operand, and min is order-free - so every matrix of minima must be equal BIT FOR BIT, in every form of the launch:
      * symmetric (triangle of tiles, persistent workgroups on the tile queues: 28 x 784 rows), whole and in 3 slices,
      * symmetric on shapes whose blocks straddle the diagonal and the image boundaries in every way (P = 100, 67, 130; K
        of 1, 2, 3, 8 stages: below one flush interval, and past it with P = 784 / D = 640: 20 stages),
      * two-way and one-way rectangles with ragged edges, one-image blocks included, and one rectangle large enough for
        the second XCD order of the new kernel - whole, and the one-way form in two slices."""
    import acnative
    from _synth import clustered_embeddings

    def both(fn):
        try:
            acnative.set_gemm_policy(acnative.GEMM_256x128)
            a = fn()
            acnative.set_gemm_policy(acnative.GEMM_AUTO)
            b = fn()
        finally:
            acnative.set_gemm_policy(acnative.GEMM_AUTO)
        return a, b

    Zs, _ = clustered_embeddings(28, 784, 128, 4, salt=58)
    pk = acnative.pack_images(torch.from_numpy(Zs).to(dev))
    a, b = both(lambda: acnative.pair_block(pk, pk))
    assert torch.isfinite(a.view(torch.float32)).all() and torch.equal(a, b)
    # (the two kernels cut the triangle into different tiles, so a slice of one holds other cells than a slice of the
    # other: what must agree is the MIN over the slices - and it must be the whole)
    wa = [both(lambda r=r: acnative.pair_block(pk, pk, part=r, nparts=3)) for r in range(3)]
    assert torch.equal(torch.stack([x[0].view(torch.float32) for x in wa]).min(0)[0],
                       torch.stack([x[1].view(torch.float32) for x in wa]).min(0)[0])
    assert torch.equal(torch.stack([x[1].view(torch.float32) for x in wa]).min(0)[0], b.view(torch.float32))
    for (n, P, D, salt) in [(9, 100, 64, 61), (7, 67, 96, 62), (11, 130, 256, 63), (3, 784, 640, 64), (1, 784, 64, 65), (5, 100, 32, 66)]:
        Zs, _ = clustered_embeddings(n, P, D, 3, salt=salt)
        pk = acnative.pack_images(torch.from_numpy(Zs).to(dev))
        a, b = both(lambda: acnative.pair_block(pk, pk))
        assert torch.equal(a, b), (n, P, D)
        for na in (1, n // 2):
            if na < 1 or na >= n:
                continue
            A, B = pk.slice(0, na), pk.slice(na, n)
            (a1, a2), (b1, b2) = both(lambda: acnative.pair_block(A, B, two_way=True))
            assert torch.equal(a1, b1) and torch.equal(a2, b2), (n, na, P, D)
            a, b = both(lambda: acnative.pair_block(B, A))
            assert torch.equal(a, b), (n, na, P, D)
    # a rectangle LARGE enough for the other XCD order of the 256x192 kernel (gemm_hl7.h: windows of 32 tiles below 160 tiles
    # per CU, one contiguous eighth per XCD above - 64 x 64 images are 196 x 262 = 51 352 tiles), two-way and one-way
    Zs, _ = clustered_embeddings(128, 784, 64, 4, salt=67)
    pk = acnative.pack_images(torch.from_numpy(Zs).to(dev))
    A, B = pk.slice(0, 64), pk.slice(64, 128)
    (a1, a2), (b1, b2) = both(lambda: acnative.pair_block(A, B, two_way=True))
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    a, b = both(lambda: acnative.pair_block(B, A))
    assert torch.equal(a, b) and torch.isfinite(a.view(torch.float32)).all()
    # ... and SLICED in two: each slice (25 676 tiles) is below the switch-over of the XCD order, the whole launch
    # above it - every part must walk the id space in the order the WHOLE launch picks, or tiles are lost between the parts
    parts = [acnative.pair_block(B, A, part=r, nparts=2) for r in range(2)]
    assert torch.equal(torch.stack([x.view(torch.float32) for x in parts]).min(0)[0], b.view(torch.float32))


def test_unknown_policy_is_rejected():
    import acnative
    with pytest.raises(RuntimeError, match="unknown policy"):
        acnative.set_gemm_policy(77)
