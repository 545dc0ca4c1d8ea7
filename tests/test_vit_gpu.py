"""GPU parity: ViT forward (patch embed, LN, attention, MLP as HIP/MFMA kernels) vs the
golden vectors captured from the reference and vs the CPU oracle on the full tensors.

Tolerance: the residual stream after 12 blocks has |x| ~ 1-10; both sides are fp32-grade
(reference: fp32 MKL; ours: split-fp16 MFMA with fp32 accumulation), so the bar is
|x - ref| <= 2e-4 + 2e-4*|ref| on every element of both tapped layers (measured ~1e-5).
"""
import numpy as np
import pytest
import torch

from _synth import gaussish_tensor, dino_like_statistics

pytestmark = pytest.mark.gpu

ATOL, RTOL = 2e-4, 2e-4


def _model(dev, depth=12):
    import patchcore.vision_transformer as vits
    torch.manual_seed(0)
    m = vits.vit_base(patch_size=8, num_classes=0, depth=depth)
    for p in m.parameters():
        p.requires_grad = False
    return m.eval().to(dev)


def _run(model, images, taps):
    x = model.prepare_tokens(images)
    out = {}
    for i, blk in enumerate(model.blocks):
        x = blk(x)
        if i in taps:
            out[i] = x.clone()
        if i == max(taps):
            break
    return out


def test_vit_base8_against_reference_golden_and_oracle(golden, dev):
    from oracle import vit as ovit
    g = golden("vit_forward.npz")
    model = _model(dev)
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=11, scale=1.7))
    got = _run(model, images.to(dev), (10, 11))
    toks = g["tokens"]
    for tap, key in ((10, "f10"), (11, "f11")):
        mine = got[tap].cpu().numpy()
        ref = g[key]
        err = np.abs(mine[:, toks] - ref)
        print("block %d: max |x-ref| on golden tokens %.3e (|ref| max %.2f)" % (tap, err.max(), np.abs(ref).max()))
        assert np.all(err <= ATOL + RTOL * np.abs(ref))
        chk = g[key + "_check"]
        assert abs(mine.astype(np.float64).sum() - chk[0]) <= 1e-5 * max(1.0, abs(chk[1]) ** 0.5) * mine.size ** 0.5
        assert abs((mine.astype(np.float64) ** 2).sum() - chk[1]) <= 1e-4 * chk[1]
    # full tensors against the oracle
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = ovit.forward_taps(sd, images, taps=(10, 11))
    for tap in (10, 11):
        mine, r = got[tap].cpu(), ref[tap]
        err = (mine - r).abs()
        print("block %d: full-tensor max err %.3e" % (tap, err.max().item()))
        assert torch.all(err <= ATOL + RTOL * r.abs())


def test_batch_not_multiple_of_anything_and_hooks(dev):
    """B=3 (token count 2355, ragged against the 128-row tiles) through the hook/early-exit API."""
    import patchcore.common as common
    from oracle import vit as ovit
    model = _model(dev, depth=2)
    images = torch.from_numpy(gaussish_tensor((3, 3, 224, 224), salt=12, scale=1.2))
    agg = common.NetworkFeatureAggregator(model, ["blocks.0", "blocks.1"], dev)
    feats = dict(agg(images.to(dev)))      # copy: the aggregator reuses (and clears) its output dict, as upstream
    assert set(feats) == {"blocks.0", "blocks.1"} and feats["blocks.1"].shape == (3, 785, 768)
    assert agg.feature_dimensions((3, 224, 224)) == [785, 785]
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = ovit.forward_taps(sd, images, taps=(0, 1))
    for i in (0, 1):
        err = (feats["blocks.%d" % i].cpu() - ref[i]).abs()
        assert torch.all(err <= ATOL + RTOL * ref[i].abs()), err.max()


def test_attention_extreme_logits(dev):
    """Force a large, late-arriving maximum in the online softmax: one key far ahead of the rest
    near the end of the sequence, plus a large common offset (rule: test the rescale branch)."""
    from oracle import vit as ovit
    model = _model(dev, depth=1)
    with torch.no_grad():
        blk = model.blocks[0]
        blk.attn.qkv.weight.mul_(40.0)          # logits ~ +-hundreds
    images = torch.from_numpy(gaussish_tensor((1, 3, 224, 224), salt=13, scale=2.0))
    got = _run(model, images.to(dev), (0,))[0].cpu()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = ovit.forward_taps(sd, images, taps=(0,))[0]
    truth = ovit.forward_taps(sd, images, taps=(0,), dtype=torch.float64)[0]
    e_ours, e_ref = (got.double() - truth).abs().max().item(), (ref.double() - truth).abs().max().item()
    print("extreme logits: |ours-truth| %.3e  |fp32 oracle-truth| %.3e  (|x| max %.1f)" % (e_ours, e_ref, truth.abs().max().item()))
    assert torch.isfinite(got).all()
    # with logits in the hundreds the softmax is a near-argmax and fp32 itself is ~1e-3 from the truth;
    # the kernel must stay within a small multiple of the fp32 reference's own distance
    assert e_ours <= max(4.0 * e_ref, 2e-3)


def test_cpu_tensor_is_refused():
    import patchcore.vision_transformer as vits
    torch.manual_seed(0)
    m = vits.vit_base(patch_size=8, num_classes=0, depth=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.prepare_tokens(torch.zeros(1, 3, 224, 224))


def test_vit_with_dino_like_statistics(golden, dev):
    """VERDICT r01 item 3c.  Trunc-normal(0.02) weights give tame activations; a trained DINO ViT-B/8 has outlier
    channels, LayerNorm gains far from 1 and non-zero biases, and the low half of the split-fp16 operands
    (lo = fp16(x - hi)) underflows for small |x| next to large ones.  Weights: the seed-0 init pushed to those
    statistics by tests/_synth.py::dino_like_statistics; golden: the REFERENCE's vit_base run on them in the
    build container (make_golden.py --dino).  Residual stream |x| up to 112 after 12 blocks; the HIP ViT must
    hold 2e-4 + 2e-4 |x| on the golden tokens and against the oracle on every element."""
    from oracle import vit as ovit
    import patchcore.vision_transformer as vits
    g = golden("vit_forward_dino_stats.npz")
    torch.manual_seed(0)
    m = vits.vit_base(patch_size=8, num_classes=0)
    for p in m.parameters():
        p.requires_grad = False
    dino_like_statistics(m.state_dict())                  # aliases the parameters: edited in place, on the CPU
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.eval().to(dev)
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=11, scale=1.7))
    got = _run(m, images.to(dev), (10, 11))
    ref = ovit.forward_taps(sd, images, taps=(10, 11))
    toks = g["tokens"]
    for tap, key in ((10, "f10"), (11, "f11")):
        mine = got[tap].cpu().numpy()
        err = np.abs(mine[:, toks] - g[key])
        print("block %d (DINO-like statistics): max |x-ref| on golden tokens %.3e, |ref| max %.1f"
              % (tap, err.max(), np.abs(g[key]).max()))
        assert np.all(err <= ATOL + RTOL * np.abs(g[key]))
        r = ref[tap].numpy()
        full = np.abs(mine - r)
        print("          vs oracle, every element: max err %.3e, max err/(atol+rtol|x|) %.3f"
              % (full.max(), (full / (ATOL + RTOL * np.abs(r))).max()))
        assert np.all(full <= ATOL + RTOL * np.abs(r))


def test_attention_ignores_the_pad_rows_of_its_buffers(dev):
    """T = 785 is not a multiple of the 32-key attention tile: the last tile reads 15 rows past the end of every
    head's K and V images.  Nothing ever writes those rows, so poison the whole scratch buffer with 0xFF bytes
    (fp16 NaNs) first: the block output must not change by a bit (padded scores are masked, padded V rows are
    re-routed to a real row)."""
    import acnative
    import patchcore.vision_transformer as vits
    torch.manual_seed(3)
    m = vits.vit_base(patch_size=8, num_classes=0, depth=1).eval().to(dev)
    x = torch.randn(3, 785, 768, device=dev)
    with torch.no_grad():
        ref = m.blocks[0](x).clone()
        for ws in acnative._workspaces.values():
            ws.fill_(0xFF)
        again = m.blocks[0](x)
    assert torch.isfinite(again).all() and torch.equal(again, ref)


def test_patch16_backbone_through_the_whole_embedding(dev, monkeypatch):
    """the other DINO names of the registry.  `dino_vitbase16` builds vit_base with 16x16 patches
    (reference backbones.py:70): 197 tokens, a 14x14 patch grid, attention tiles with 5 real keys in the last tile.
    Taps of blocks 10/11 and the aggregated Z (3x3 unfold on the 14x14 grid, 6912 -> 2048 per layer - the register
    path of the aggregation kernel - concat to 4096) against the oracle."""
    import patchcore.backbones
    import patchcore.common as common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from oracle import vit as ovit, aggregate as oagg
    monkeypatch.setenv("AC_BACKBONE_SEED", "4")
    model = patchcore.backbones.load("dino_vitbase16")
    assert model.pos_embed.shape == (1, 197, 768) and model.weights_source == "random-init(seed=4)"
    core = AnomalyClusteringCore(dev)
    core.load(backbone=model, layers_to_extract_from=["blocks.10", "blocks.11"], device=dev, input_shape=(3, 224, 224),
              pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), nn_method=common.FaissNN(False, 4))
    images = torch.from_numpy(gaussish_tensor((3, 3, 224, 224), salt=14, scale=1.5))
    Z = core.embed_device(images.to(dev), batch_size=2)               # two batches: in-place rows of one Z
    assert Z.shape == (3, 196, 4096)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    taps = ovit.forward_taps(sd, images, taps=(10, 11))
    feats = core.forward_modules["feature_aggregator"](images.to(dev))
    for i in (10, 11):
        r = taps[i]
        err = (feats["blocks.%d" % i].cpu() - r).abs()
        assert torch.all(err <= ATOL + RTOL * r.abs()), (i, err.max())
    Z_ref = oagg.aggregate([taps[10], taps[11]], 3, 2048, 4096)[0].reshape(3, 196, 4096)
    zerr = (Z.cpu() - Z_ref).abs()
    print("patch 16: Z max err %.2e" % zerr.max().item())
    assert torch.all(zerr <= 2e-4 + 2e-4 * Z_ref.abs())


@pytest.mark.parametrize("name,width,heads,depth", [("vit_small", 384, 6, 3), ("vit_large", 1024, 16, 2)])
def test_registered_timm_widths_against_oracle(dev, name, width, heads, depth):
    """`backbones.load` offers timm's vit_small (C = 384, 6 heads) and vit_large (C = 1024, 16 heads,
    hidden 4096) (reference backbones.py:34-38), but only C = 768 ran on the GPU.  The ViT kernels at those widths
    (depth cut to keep the CPU oracle fast: the blocks are identical), two images, every element of the last two taps
    against the oracle - the gate of test_vit_base8_against_reference_golden_and_oracle - and `_embed` through the
    aggregator (taps of width C -> 2 x 2048 -> 4096) against the oracle's aggregation."""
    import patchcore.backbones as backbones
    import patchcore.common as common
    import patchcore.sampler
    import patchcore.vision_transformer as vits
    from patchcore.patchcore import AnomalyClusteringCore
    from oracle import vit as ovit, aggregate as oagg
    timm_name, patch, w, full_depth, h = backbones._TIMM_VIT[name]
    assert (w, h, patch) == (width, heads, 8)
    torch.manual_seed(0)
    model = vits._vit(width, heads, patch, depth=depth, num_classes=0)
    for p in model.parameters():
        p.requires_grad = False
    model = model.eval().to(dev)
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=21, scale=1.5))
    taps = (depth - 2, depth - 1)
    got = _run(model, images.to(dev), taps)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = ovit.forward_taps(sd, images, taps=taps, num_heads=heads)
    for tap in taps:
        mine, r = got[tap].cpu(), ref[tap]
        assert mine.shape == (2, 785, width)
        err = (mine - r).abs()
        print("%s block %d: full-tensor max err %.3e (|x| max %.2f)" % (name, tap, err.max().item(), r.abs().max().item()))
        assert torch.all(err <= ATOL + RTOL * r.abs())
    core = AnomalyClusteringCore(dev)
    core.load(backbone=model, layers_to_extract_from=["blocks.%d" % t for t in taps], device=dev, input_shape=(3, 224, 224),
              pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), nn_method=common.FaissNN(False, 4))
    Z = core.embed_device(images.to(dev), batch_size=2).cpu()
    Zref = oagg.aggregate([ref[taps[0]], ref[taps[1]]], 3, 2048, 4096)[0].reshape(2, 784, 4096)
    errz = (Z - Zref).abs()
    print("%s Z: max err %.3e" % (name, errz.max().item()))
    assert torch.all(errz <= 3e-5 + 2e-5 * Zref.abs())
