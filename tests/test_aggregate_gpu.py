"""GPU parity: fused patch-aggregation kernel vs vectors captured from the reference's
AnomalyClusteringCore._embed (tests/golden/aggregate_*.npz) and vs the oracle.

Tolerance: Z is a mean of 4-5 standardised fp32 values (|Z| <~ 4): |Z - ref| <= 3e-6 + 2e-6*|ref|
(differences come only from the mean/rstd rounding of the per-image standardisation).
"""
import numpy as np
import pytest
import torch

from _synth import gaussish_tensor

pytestmark = pytest.mark.gpu

ATOL, RTOL = 3e-6, 2e-6


def _close(a, ref):
    err = np.abs(a - ref)
    ok = np.all(err <= ATOL + RTOL * np.abs(ref))
    return ok, err.max()


@pytest.mark.parametrize("tag,salts,shape_a,shape_b,off", [
    ("tok_small", (21, 22), (2, 37, 64), (2, 37, 64), (0.3, -0.1)),
    ("tok_p5", (21, 22), (2, 37, 64), (2, 37, 64), (0.3, -0.1)),
    ("cnn_small", (23, 24), (2, 32, 12, 12), (2, 48, 6, 6), (0.0, 0.0)),
])
def test_variants_against_reference_golden(golden, dev, tag, salts, shape_a, shape_b, off):
    import acnative
    g = golden("aggregate_%s.npz" % tag)
    scales = {"tok_small": (2.0, 1.0), "tok_p5": (2.0, 1.0), "cnn_small": (2.0, 1.5)}[tag]
    fa = torch.from_numpy(gaussish_tensor(shape_a, salt=salts[0], scale=scales[0])) + off[0]
    fb = torch.from_numpy(gaussish_tensor(shape_b, salt=salts[1], scale=scales[1])) + off[1]
    Z, grids = acnative.patch_aggregate([fa.to(dev), fb.to(dev)], int(g["patchsize"]), int(g["pre"]), int(g["tgt"]))
    ok, mx = _close(Z.cpu().numpy(), g["Z"])
    print(tag, "max err %.3e" % mx, grids)
    assert Z.shape == g["Z"].shape and ok


def test_wrn50_shaped_cnn_branch(golden, dev):
    """BASELINE config 1's shapes: layer2 [512,28,28] + layer3 [1024,14,14] (bilinear 14->28), 1024->1024."""
    import acnative
    g = golden("aggregate_wrn.npz")
    wa = torch.from_numpy(gaussish_tensor((1, 512, 28, 28), salt=25, scale=2.0))
    wb = torch.from_numpy(gaussish_tensor((1, 1024, 14, 14), salt=26, scale=1.0))
    Z, grids = acnative.patch_aggregate([wa.to(dev), wb.to(dev)], 3, 1024, 1024)
    Zc = Z.cpu().numpy()
    ok, mx = _close(Zc[g["rows"]], g["Z_rows"])
    print("wrn max err %.3e" % mx)
    assert ok and grids == [[28, 28], [14, 14]]
    np.testing.assert_allclose(Zc.astype(np.float64).sum(1), g["Z_rowsum"], rtol=0, atol=2e-3)


def test_vit_shaped_against_oracle_full(dev):
    """[B,785,768] x 2 layers, 3x3, 2048 -> 4096: the headline configuration, every element."""
    import acnative
    from oracle import aggregate as oagg
    f10 = torch.from_numpy(gaussish_tensor((2, 785, 768), salt=81, scale=3.0)) + 0.2
    f11 = torch.from_numpy(gaussish_tensor((2, 785, 768), salt=82, scale=5.0)) - 0.4
    ref, rgrids = oagg.aggregate([f10, f11], 3, 2048, 4096)
    Z, grids = acnative.patch_aggregate([f10.to(dev), f11.to(dev)], 3, 2048, 4096)
    ok, mx = _close(Z.cpu().numpy(), ref.numpy())
    print("vit-shaped max err %.3e" % mx)
    assert ok and grids == rgrids and Z.shape == (2 * 784, 4096)


def test_standalone_unfold_and_pool_vs_oracle(dev):
    import acnative
    from oracle import aggregate as oagg
    x = torch.from_numpy(gaussish_tensor((2, 5, 7, 6), salt=83))
    ref, grid = oagg.patchify(x, 3)
    got = acnative.unfold_patches(x.to(dev), 3).cpu()
    assert torch.equal(got, ref.contiguous())
    rows = torch.from_numpy(gaussish_tensor((9, 6912), salt=84))
    for out in (2048, 1000, 6912, 7000):
        ref = oagg.mean_map(rows, out)
        got = acnative.adaptive_pool_rows(rows.to(dev), out).cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)


def test_adaptive_windows_match_survey():
    from oracle.aggregate import adaptive_windows
    s, e = adaptive_windows(6912, 2048)
    widths = e - s
    assert (widths == 4).sum() == 1536 and (widths == 5).sum() == 512
    s2, e2 = adaptive_windows(4096, 4096)
    assert np.array_equal(e2 - s2, np.ones(4096, dtype=np.int64))          # identity at 2*2048 -> 4096


def test_register_path_equals_generic_path(dev):
    """The DINO ViT shape (token maps, C = 768, 3x3, 6912 -> 2048 per layer, identity final pool) runs the register
    kernel (no LDS); the same feature maps handed over as NCHW run the generic LDS kernel.  Same standardisation,
    same pooling windows summed in the same order: equal up to the last bit of the per-image statistics (their fp64
    partial sums are grouped by memory layout)."""
    import acnative
    B = 3
    toks = [torch.from_numpy(gaussish_tensor((B, 785, 768), salt=81 + l, scale=2.0 + l)).to(dev) + 0.25 * l for l in range(2)]
    Z_reg, grids = acnative.patch_aggregate(toks, 3, 2048, 4096)
    nchw = [t[:, 1:].reshape(B, 28, 28, 768).permute(0, 3, 1, 2).contiguous() for t in toks]
    Z_gen, _ = acnative.patch_aggregate(nchw, 3, 2048, 4096)
    assert grids == [[28, 28], [28, 28]] and Z_reg.shape == (B * 784, 4096)
    a, b = Z_reg.cpu().numpy(), Z_gen.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-6)
    assert (a == b).mean() > 0.9
    # one layer, and a shape that must NOT take the register path (pre = 1024): still the LDS kernels' answer
    Z1, _ = acnative.patch_aggregate(toks[:1], 3, 2048, 2048)
    np.testing.assert_allclose(Z1.cpu().numpy(), a[:, :2048], rtol=2e-6, atol=1e-6)


def test_row_kernel_equals_the_per_patch_kernel_bitwise(dev):
    """Round 4: batches of >= 8 row-workgroups per CU (here 80 images of the DINO ViT-B/8 shape) run the register kernel
    with one workgroup per ROW of patches (a sliding 3 x 3 window of standardised tokens); smaller batches keep one
    workgroup per patch.  The same images through both - one call of 80, four calls of 20 - give the SAME BITS, and three
    rows of one image (first, interior, last: the zero padding on every side) match the oracle."""
    import acnative
    from oracle import aggregate as oagg
    fa = torch.from_numpy(gaussish_tensor((80, 785, 768), salt=31, scale=1.4)) + 0.2
    fb = torch.from_numpy(gaussish_tensor((80, 785, 768), salt=32, scale=0.9)) - 0.1
    fa, fb = fa.to(dev), fb.to(dev)
    Z_row, _ = acnative.patch_aggregate([fa, fb], 3, 2048, 4096)
    Z_patch = torch.cat([acnative.patch_aggregate([fa[i:i + 20].contiguous(), fb[i:i + 20].contiguous()], 3, 2048, 4096)[0]
                         for i in range(0, 80, 20)])
    assert Z_row.shape == (80 * 784, 4096) and torch.equal(Z_row, Z_patch)
    ref = oagg.aggregate([fa[7:8].cpu(), fb[7:8].cpu()], 3, 2048, 4096)[0].numpy()
    mine = Z_row[7 * 784: 8 * 784].cpu().numpy()
    for rows in (slice(0, 28), slice(13 * 28, 14 * 28), slice(27 * 28, 28 * 28)):
        ok, mx = _close(mine[rows], ref[rows])
        assert ok, mx


def test_hl32_rows_straight_from_the_aggregation_equal_pack_of_Z_bitwise(dev):
    """Round 6: ac_patch_aggregate_hl32 writes the operand rows of the distance kernels itself (the shards a rank receives over
    the ring: their fp32 Z need not exist).  Byte for byte ac_pack_hl32 of ac_patch_aggregate's Z - through the row kernel (80
    images) and through the per-patch kernel (20 images) - and refused, with the support query saying so, for every shape
    off the register path (a CNN map, pre_dim 1024, patch size 5)."""
    import acnative
    fa = torch.from_numpy(gaussish_tensor((80, 785, 768), salt=31, scale=1.4)) + 0.2
    fb = torch.from_numpy(gaussish_tensor((80, 785, 768), salt=32, scale=0.9)) - 0.1
    fa, fb = fa.to(dev), fb.to(dev)
    for n in (80, 20):
        feats = [fa[:n].contiguous(), fb[:n].contiguous()]
        assert acnative.patch_aggregate_hl32_supported(feats, 3, 2048, 4096)
        Z, _ = acnative.patch_aggregate(feats, 3, 2048, 4096)
        ref = acnative.pack_images(Z.reshape(n, 784, 4096))
        hl = torch.full((n * 784, 4096 * 4), 0xAB, dtype=torch.uint8, device=dev)
        acnative.patch_aggregate_hl32(feats, 3, 2048, 4096, hl)
        assert torch.equal(hl, ref.hl), n
    cnn = [torch.zeros((2, 512, 28, 28), device=dev), torch.zeros((2, 1024, 14, 14), device=dev)]
    assert not acnative.patch_aggregate_hl32_supported(cnn, 3, 1024, 1024)
    assert not acnative.patch_aggregate_hl32_supported([fa[:2].contiguous()], 3, 1024, 1024)
    assert not acnative.patch_aggregate_hl32_supported([fa[:2].contiguous(), fb[:2].contiguous()], 5, 2048, 4096)
    with pytest.raises(RuntimeError, match="register path"):
        acnative.patch_aggregate_hl32([fa[:2].contiguous()], 3, 1024, 1024, torch.empty((2 * 784, 1024 * 4), dtype=torch.uint8, device=dev))
