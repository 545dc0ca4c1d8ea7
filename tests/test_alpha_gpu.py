"""GPU parity: alpha path (pair-min-distance MFMA kernel, fp64 softmax, weighted average,
pdist) through the C ABI vs the oracle and the golden vectors captured from the reference.

Tolerances (stated once, used below):
  w      : |w - w_ref| <= 3e-5 + 2e-6*|w_ref|   (both sides are fp32-grade; the reference's own
           fp32 cdist is ~1e-5 from the fp64 truth at w ~ 30)
  alpha  : relative <= 2e-4 for tau >= 0.2 (alpha amplifies dw by 1/tau), absolute <= 1e-6
  X      : |X - X_ref| <= 2e-5 (fp32 accumulation-order differences over 784 terms)
  pdist  : relative <= 1e-12 (both fp64)
"""
import numpy as np
import pytest
import torch

from _synth import clustered_embeddings, gaussish_tensor

pytestmark = pytest.mark.gpu

W_ATOL, W_RTOL = 3e-5, 2e-6
A_RTOL, A_ATOL = 2e-4, 1e-6


def _w_close(w, ref):
    return np.all(np.abs(w - ref) <= W_ATOL + W_RTOL * np.abs(ref))


def _truth_w(Zq, Zk, reduce, self_offset=0):
    """float64 brute force."""
    Zq, Zk = Zq.astype(np.float64), Zk.astype(np.float64)
    out = np.zeros(Zq.shape[:2])
    for i in range(Zq.shape[0]):
        cols = []
        for j in range(Zk.shape[0]):
            if reduce == 0 and j == i + self_offset:
                continue
            d2 = (Zq[i] ** 2).sum(1)[:, None] + (Zk[j] ** 2).sum(1)[None, :] - 2 * Zq[i] @ Zk[j].T
            cols.append(np.sqrt(np.maximum(d2, 0)).min(1))
        cols = np.stack(cols, 1)
        out[i] = cols.mean(1) if reduce == 0 else cols.min(1)
    return out


def test_hl32_split_is_the_two_fp16_roundings_bit_for_bit(dev):
    """hl32 = (hi, lo) with hi = fp16(x), lo = fp16(x - hi).  The device forms lo with
    v_fma_mix{lo,hi}_f16 (common.h split2: hi * (-1) + x in fp32, one rounding to fp16); x - float(hi) is exact in fp32, so the
    planes must equal the two plain roundings BIT FOR BIT - on ordinary values, on the fp16 boundaries (subnormal lo halves,
    the largest finite hi, values that round to inf) and on inf / nan."""
    import acnative
    g = torch.Generator().manual_seed(5)
    x = torch.randn((64, 256), generator=g)
    x[1] *= 1e-3
    x[2] *= 1e3
    x[3] *= 6e-5                     # fp16 subnormal hi halves
    x[4] *= 1e-7                     # lo halves (and hi halves) far below the smallest fp16 subnormal
    x[5] = torch.linspace(65400.0, 65540.0, 256)      # across the largest finite fp16 (65504) and the rounding boundary 65520
    x[6, :8] = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 65504.0, 65519.996, 2.0 ** -24])
    x[7] = torch.arange(256, dtype=torch.float32) * 2.0 ** -14 + 1.0     # ties of the hi rounding (round to nearest even)
    hl = acnative.pack_hl32(x.to(dev)).cpu().reshape(64, 256 // 32, 2, 32, 2)         # [row][group][plane][k][byte]
    got_hi = hl[:, :, 0].reshape(64, 256, 2).contiguous().view(torch.float16).reshape(64, 256)
    got_lo = hl[:, :, 1].reshape(64, 256, 2).contiguous().view(torch.float16).reshape(64, 256)
    hi = x.to(torch.float16)
    lo = (x - hi.to(torch.float32)).to(torch.float16)
    for got, ref, name in ((got_hi, hi, "hi"), (got_lo, lo, "lo")):
        nan = torch.isnan(ref)
        assert torch.equal(torch.isnan(got), nan), name
        assert torch.equal(got.view(torch.int16)[~nan], ref.view(torch.int16)[~nan]), name


def test_small_all_taus_against_reference_golden(golden, dev):
    import acnative
    g = golden("alpha_small.npz")
    Zs, _ = clustered_embeddings(6, 49, 128, 3, salt=31)
    Zt, _ = clustered_embeddings(4, 49, 128, 1, salt=32)
    Z, Ztr = torch.from_numpy(Zs).to(dev), torch.from_numpy(Zt).to(dev)
    w_u = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS)
    w_s = acnative.pair_min_dist(Z, Ztr, acnative.REDUCE_MIN_ALL)
    assert _w_close(w_u.cpu().numpy(), g["w_unsup"])
    assert _w_close(w_s.cpu().numpy(), g["w_sup"])
    for tau in g["taus"]:
        for mode, w in (("unsup", w_u), ("sup", w_s)):
            ref = g["alpha_%s_%g" % (mode, tau)]
            a = acnative.alpha_softmax(w, float(tau), 1.0).cpu().numpy()
            assert a.dtype == np.float64 and a.shape == ref.shape
            np.testing.assert_allclose(a.sum(1), 1.0, rtol=1e-12)
            if tau == 0:
                assert np.array_equal(a > 0, ref > 0), "arg-max indicator differs"
                np.testing.assert_allclose(a, ref, rtol=0, atol=1e-15)
            else:
                np.testing.assert_allclose(a, ref, rtol=A_RTOL / min(1.0, tau / 0.2), atol=A_ATOL)


def test_softmax_exact_on_reference_weights(golden, dev):
    """Feed the reference's own w: the fp64 softmax must then agree to ~1e-15."""
    import acnative
    g = golden("alpha_small.npz")
    w = torch.from_numpy(g["w_unsup"]).to(dev)
    for tau in g["taus"]:
        a = acnative.alpha_softmax(w, float(tau), 1.0).cpu().numpy()
        np.testing.assert_allclose(a, g["alpha_unsup_%g" % tau], rtol=1e-12, atol=1e-300)


def test_tau_zero_ties(golden, dev):
    import acnative
    g = golden("alpha_small.npz")
    Zs, _ = clustered_embeddings(6, 49, 128, 3, salt=31)
    Zs[0, 5] = Zs[0, 9]
    Zs[0, 5] += 40.0
    Zs[0, 9] = Zs[0, 5]
    Z = torch.from_numpy(Zs).to(dev)
    w = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS)
    a = acnative.alpha_softmax(w, 0.0, 1.0).cpu().numpy()
    assert (a[0] == 0.5).sum() == 2 and a[0, 5] == 0.5 and a[0, 9] == 0.5
    np.testing.assert_array_equal(a, g["alpha_tie_unsup_0"])


def test_full_size_against_reference_golden(golden, dev):
    """P=784, D=4096.  At this depth the reference's own fp32 cdist (norms + -2ab accumulated
    in one fp32 GEMM) sits ~1e-4 from the fp64 truth; the split-fp16 MFMA path is ~10x closer.
    So: tight against the truth, and against the reference no further than the reference's
    own distance from the truth."""
    import acnative
    g = golden("alpha_full.npz")
    Zf, _ = clustered_embeddings(3, 784, 4096, 3, salt=41)
    Zb, _ = clustered_embeddings(2, 784, 4096, 1, salt=42)
    Z, Zbk = torch.from_numpy(Zf).to(dev), torch.from_numpy(Zb).to(dev)
    w_u = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS).cpu().numpy()
    w_s = acnative.pair_min_dist(Z, Zbk, acnative.REDUCE_MIN_ALL).cpu().numpy()
    t_u, t_s = _truth_w(Zf, Zf, 0), _truth_w(Zf, Zb, 1)
    e_ours = max(np.abs(w_u - t_u).max(), np.abs(w_s - t_s).max())
    e_ref = max(np.abs(g["w_unsup"] - t_u).max(), np.abs(g["w_sup"] - t_s).max())
    e_pair = max(np.abs(w_u - g["w_unsup"]).max(), np.abs(w_s - g["w_sup"]).max())
    print("w: |ours-truth| %.3e  |ref-truth| %.3e  |ours-ref| %.3e" % (e_ours, e_ref, e_pair))
    assert e_ours <= 2e-5
    assert e_pair <= e_ref + 2e-5 and e_pair <= 5e-4
    wd = torch.from_numpy(w_u).to(dev)
    for tau, key in ((1.0, "alpha_unsup_1"), (0.2, "alpha_unsup_0.2")):
        a = acnative.alpha_softmax(wd, tau).cpu().numpy()
        z = (t_u - t_u.max(1, keepdims=True)) / tau
        a_truth = np.exp(z) / np.exp(z).sum(1, keepdims=True)
        np.testing.assert_allclose(a, a_truth, rtol=A_RTOL, atol=A_ATOL)          # vs fp64 truth
        np.testing.assert_allclose(a, g[key], rtol=5e-4 / tau, atol=A_ATOL)       # vs reference (its own w error / tau)
    X = acnative.weighted_avg(torch.from_numpy(g["alpha_unsup_1"]).float().to(dev), Z).cpu().numpy()
    np.testing.assert_allclose(X, g["X_unsup_1"], rtol=0, atol=2e-5)


def test_query_shard_matches_whole(dev):
    """Sharded queries with self_offset (the multi-GPU call shape) vs the unsharded call.  The unsharded
    call takes the symmetric path (only tiles tm <= tn, each reduced in both directions); the sharded /
    cloned-key calls take the general path: same values up to the order of two fp32 additions."""
    import acnative
    Zs, _ = clustered_embeddings(7, 100, 256, 3, salt=51)
    Z = torch.from_numpy(Zs).to(dev)
    whole = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS)
    general = acnative.pair_min_dist(Z, Z.clone(), acnative.REDUCE_MEAN_OTHERS)
    staged = acnative.pair_min_dist_staged(Z, Z, acnative.REDUCE_MEAN_OTHERS)
    part = acnative.pair_min_dist(Z[3:6].contiguous(), Z, acnative.REDUCE_MEAN_OTHERS, self_offset=3)
    part_staged = acnative.pair_min_dist_staged(Z[3:6], Z, acnative.REDUCE_MEAN_OTHERS, self_offset=3)
    assert torch.equal(general[3:6], part) and torch.equal(part, part_staged)      # general path: bit for bit
    assert torch.equal(whole, staged)                                               # symmetric path: bit for bit
    np.testing.assert_allclose(whole.cpu().numpy(), general.cpu().numpy(), rtol=1e-6, atol=1e-6)
    ref = _truth_w(Zs, Zs, 0)
    assert np.abs(whole.cpu().numpy() - ref).max() <= 2e-5
    assert np.abs(general.cpu().numpy() - ref).max() <= 2e-5


def test_partitioned_symmetric_slices_combine_to_the_whole(dev):
    """What rank r of G computes in the multi-GPU unsupervised path: slice r of the upper-triangle
    tiles into its own minbuf.  The element-wise MIN over the G slices (the all-reduce) must equal
    the single-launch matrix bit for bit, for G that do and do not divide the tile count."""
    import acnative
    Zs, _ = clustered_embeddings(5, 784, 256, 3, salt=55)          # 3920 rows: 16 x 31 tiles, ragged edge
    Z = torch.from_numpy(Zs).to(dev)
    whole = acnative.pair_min_sq_part(Z, 0, 1)
    assert torch.isfinite(whole).all()
    for G in (2, 3, 8):
        parts = [acnative.pair_min_sq_part(Z, r, G) for r in range(G)]
        combined = torch.stack(parts).min(0)[0]
        assert torch.equal(combined, whole), G
        assert sum(int(torch.isfinite(p).sum()) for p in parts) >= whole.numel()     # slices cover everything
    msq = acnative.max_sq([acnative.pack_images(Z)], dev)          # the category's fixed-point scale
    w = acnative.pair_reduce(whole, 784, acnative.REDUCE_MEAN_OTHERS, 0, max_sq_t=msq)
    assert torch.equal(w, acnative.pair_min_dist_staged(Z, Z, acnative.REDUCE_MEAN_OTHERS))
    assert torch.equal(w, acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS))
    w_part = acnative.pair_reduce(whole[:, 784:3 * 784], 784, acnative.REDUCE_MEAN_OTHERS, 1, max_sq_t=msq, n_keys=5)
    assert torch.equal(w_part, w[1:3])


def test_dynamic_tile_queues_on_sliced_and_repeated_launches(dev):
    """Round 4: launches of >= 4 tiles per CU walk their tiles through per-XCD-slot queues with stealing (gemm_hl2.h).
    28 images x 784 patches = 86 x 172 tiles of the triangle: the whole launch and its slices (id_lo != 0, slice
    boundaries at multiples of 8) all take the dynamic walk.  The MIN over the slices equals the whole, bit for bit; the
    whole equals the brute-force float64 minima on a sample of cells; and 70 back-to-back launches (more than the 64
    counter sets that rotate) keep giving the same bits - every launch finds its counters zeroed by the last workgroup of
    the launch that used them before."""
    import acnative
    Zs, _ = clustered_embeddings(28, 784, 128, 4, salt=58)
    Z = torch.from_numpy(Zs).to(dev)
    whole = acnative.pair_min_sq_part(Z, 0, 1)
    assert torch.isfinite(whole).all()
    for G in (2, 3):
        parts = [acnative.pair_min_sq_part(Z, r, G) for r in range(G)]
        assert torch.equal(torch.stack(parts).min(0)[0], whole), G
    Zd = Z.double()
    for (j, i) in [(0, 1), (5, 27), (27, 3), (13, 13)]:            # minbuf[j][i*P + p] = min_q |Z[i][p] - Z[j][q]|^2
        d2 = torch.cdist(Zd[i], Zd[j]).pow(2).min(1)[0]
        got = whole[j, i * 784:(i + 1) * 784].double()
        if i == j:
            assert float(got.max()) <= 5e-3                          # a patch is its own nearest neighbour (|z|^2 ~ 140: fp32-grade zero)
        else:
            assert float((got - d2).abs().max()) <= 2e-5 * float(d2.max()) + 1e-6
    pk = acnative.pack_images(Z)
    for _ in range(70):
        again = acnative.pair_block(pk, pk).view(torch.float32)
    assert torch.equal(again, whole)


def test_two_way_block_against_one_way_blocks(dev):
    """The two-way launch (a rectangle between two different image sets, every tile reduced in both
    directions): the forward matrix is exactly what a one-way launch fills; the transposed matrix holds the
    same distances as the one-way launch with the operands exchanged, up to the last bits (a.b and b.a
    accumulate their three split-fp16 passes in a different order) - ragged shapes: P not a multiple of the
    tile, nk != nq, blocks of one image."""
    import acnative
    for (na, nb, P, D, salt) in [(3, 2, 100, 256, 57), (1, 4, 67, 96, 58), (2, 2, 784, 128, 59), (5, 1, 130, 64, 60)]:
        Zs, _ = clustered_embeddings(na + nb, P, D, 3, salt=salt)
        pk = acnative.pack_images(torch.from_numpy(Zs).to(dev))
        A, B = pk.slice(0, na), pk.slice(na, na + nb)
        mb, mbt = acnative.pair_block(A, B, two_way=True)
        assert torch.equal(mb, acnative.pair_block(A, B)), (na, nb, P, D)
        other = acnative.pair_block(B, A).view(torch.float32)
        assert (mb != acnative.INF_BITS).all() and (mbt != acnative.INF_BITS).all()
        np.testing.assert_allclose(mbt.view(torch.float32).cpu().numpy(), other.cpu().numpy(), rtol=2e-6, atol=1e-6)
    with pytest.raises(ValueError, match="symmetric case"):
        acnative.pair_block(A, A, two_way=True)


class _IdentityOps:
    """HipOps whose 'images' ARE the embeddings Z [N, P, D]: isolates the alpha engine from the ViT."""

    def __new__(cls):
        from patchcore.pipeline import HipOps
        ops = HipOps(core=None)
        ops.embed = lambda images: images.contiguous()
        return ops


def test_block_schedules_are_bit_identical_on_one_gpu(dev):
    """run_category's unsupervised engine on one GPU: one symmetric launch (default budget) vs the ring
    schedule forced into sub-blocks of 1, 2 and 3 images (symmetric diagonal blocks + two-way off-diagonal
    blocks + the fixed-point sum): the same bits.  Supervised: sub-blocked one-way blocks == one launch."""
    from patchcore.pipeline import run_category
    from oracle import alpha as oa
    import acnative
    Zs, _ = clustered_embeddings(7, 100, 256, 3, salt=51)
    Zb, _ = clustered_embeddings(5, 100, 256, 1, salt=52)
    Z, Zbank = torch.from_numpy(Zs).to(dev), torch.from_numpy(Zb).to(dev)
    ops = _IdentityOps()
    ref = run_category(ops, Z, "unsupervised", tau=1.0)
    assert torch.equal(ref["w_loc"], acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS))
    assert _w_close(ref["w_loc"].cpu().numpy(), oa.weights_unsupervised(torch.from_numpy(Zs)).numpy())
    for c in (1, 2, 3):
        got = run_category(ops, Z, "unsupervised", tau=1.0, budget=c * c * 100 * 4)
        assert torch.equal(got["w_loc"], ref["w_loc"]), c
        assert torch.equal(got["alpha"], ref["alpha"]) and np.array_equal(got["X"], ref["X"])
    sup = run_category(ops, Z, "supervised", tau=1.0, train_images=Zbank, train_ratio=1.0)
    sup_small = run_category(ops, Z, "supervised", tau=1.0, train_images=Zbank, train_ratio=1.0, budget=2 * 2 * 100 * 4)
    assert torch.equal(sup["w_loc"], sup_small["w_loc"])
    assert _w_close(sup["w_loc"].cpu().numpy(),
                    oa.weights_supervised(torch.from_numpy(Zs), torch.from_numpy(Zb)).numpy())


def _split_values(Z):
    """What the kernels see of Z: hi + lo of the split-fp16 representation, as float64."""
    hi = Z.to(torch.float16)
    lo = (Z - hi.float()).to(torch.float16)
    return hi.double() + lo.double()


def test_pruned_nearest_neighbour_is_exact(dev):
    """Supervised weights as a pruned search (fp16 first pass + exact re-measurement of the survivors) against
    (a) the brute force over ALL bank patches in float64 on the same split values - the pruning must never lose the
    nearest neighbour, so the two agree to fp64 round-off (1e-6 relative after the sqrt and the cast) - and
    (b) the exhaustive split-fp16 path (2e-5, its own distance from the truth)."""
    from patchcore.pipeline import run_category
    ops = _IdentityOps()
    for (nq, nk, P, D, salt) in [(4, 5, 100, 256, 71), (2, 3, 784, 4096, 72), (3, 1, 67, 64, 73)]:
        Zq, _ = clustered_embeddings(nq, P, D, 3, salt=salt)
        Zk, _ = clustered_embeddings(nk, P, D, 1, salt=salt + 100)
        tq, tk = torch.from_numpy(Zq).to(dev), torch.from_numpy(Zk).to(dev)
        pruned = run_category(ops, tq, "supervised", tau=1.0, train_images=tk, train_ratio=float(nk) / nq + 1)["w_loc"]
        exhaustive = run_category(ops, tq, "supervised", tau=1.0, train_images=tk, train_ratio=float(nk) / nq + 1,
                                  prune=False)["w_loc"]
        a, b = _split_values(tq).reshape(nq * P, D), _split_values(tk).reshape(nk * P, D)
        d2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
        brute = d2.clamp_min(0).min(1)[0].sqrt().reshape(nq, P)
        np.testing.assert_allclose(pruned.double().cpu().numpy(), brute.cpu().numpy(), rtol=1e-6, atol=1e-6)
        assert (pruned - exhaustive).abs().max().item() <= 2e-5, (nq, nk, P, D)


def test_pruned_search_re_measures_queries_whose_candidate_lists_overflow(dev):
    """A bank made of copies of one image gives every query patch many equally near bank patches; with one candidate
    slot the lists overflow and those queries go through the exhaustive kernel on a compact copy: their answer is
    the exhaustive path's, bit for bit; with room for the ties the pruned answer agrees to 2e-5."""
    from patchcore.pipeline import run_category
    Zq, _ = clustered_embeddings(3, 100, 128, 2, salt=81)
    one, _ = clustered_embeddings(1, 100, 128, 1, salt=82)
    bank = np.repeat(one, 6, axis=0)
    tq, tk = torch.from_numpy(Zq).to(dev), torch.from_numpy(bank).to(dev)
    ops = _IdentityOps()
    ops.nn_cap = 1
    tight = run_category(ops, tq, "supervised", tau=1.0, train_images=tk, train_ratio=2.0)["w_loc"]
    exhaustive = run_category(_IdentityOps(), tq, "supervised", tau=1.0, train_images=tk, train_ratio=2.0, prune=False)["w_loc"]
    assert torch.equal(tight, exhaustive)                       # every query has >= 6 ties: all of them overflowed
    roomy = run_category(_IdentityOps(), tq, "supervised", tau=1.0, train_images=tk, train_ratio=2.0)["w_loc"]
    assert (roomy - exhaustive).abs().max().item() <= 2e-5
    # a mixed case: two slots, a bank with one duplicated image - some queries overflow, some do not
    mixed_bank = np.concatenate([one, one, clustered_embeddings(2, 100, 128, 1, salt=83)[0]])
    ops2 = _IdentityOps()
    ops2.nn_cap = 2
    tm = torch.from_numpy(mixed_bank).to(dev)
    got = run_category(ops2, tq, "supervised", tau=1.0, train_images=tm, train_ratio=2.0)["w_loc"]
    ref = run_category(_IdentityOps(), tq, "supervised", tau=1.0, train_images=tm, train_ratio=2.0, prune=False)["w_loc"]
    assert (got - ref).abs().max().item() <= 2e-5


def test_pruned_search_leaves_norms_beyond_fp16_to_the_exhaustive_kernel(dev):
    """The error bound of the first pass needs finite fp16 high halves, which |row| < 65504 guarantees.  Rows with a
    larger norm (here every element is still inside fp16's range: D = 256 values of ~5000 give |row| ~ 80 000) flip
    ac_nn_slack's range flag: every query is treated as overflowed and the exhaustive split-fp16 blocks decide - the
    answer is prune=False's, bit for bit, and finite (the pruned path used to return silently wrong
    minima once a high half became inf)."""
    import acnative
    from patchcore.pipeline import run_category
    Zq, _ = clustered_embeddings(3, 100, 256, 2, salt=85)
    Zk, _ = clustered_embeddings(4, 100, 256, 1, salt=86)
    tq, tk = torch.from_numpy(Zq * 8000.0).to(dev), torch.from_numpy(Zk * 8000.0).to(dev)
    pk = acnative.pack_images(tk)
    assert float(acnative.max_sq([pk], dev).sqrt().item()) > 65504.0 and float(tk.abs().max().item()) < 65504.0
    s = acnative.nn_slack(acnative.max_sq([pk], dev), acnative.max_sq([acnative.pack_images(tq)], dev))
    assert float(s[2].item()) == 1.0
    pruned = run_category(_IdentityOps(), tq, "supervised", tau=1.0, train_images=tk, train_ratio=2.0)["w_loc"]
    exhaustive = run_category(_IdentityOps(), tq, "supervised", tau=1.0, train_images=tk, train_ratio=2.0, prune=False)["w_loc"]
    assert torch.isfinite(pruned).all() and torch.equal(pruned, exhaustive)
    # in range: the flag stays clear
    s = acnative.nn_slack(acnative.max_sq([acnative.pack_images(torch.from_numpy(Zk).to(dev))], dev),
                          acnative.max_sq([acnative.pack_images(torch.from_numpy(Zq).to(dev))], dev))
    assert float(s[2].item()) == 0.0


def test_fixed_point_scale_follows_the_data(dev):
    """The shift of the 64-bit fixed-point sum comes from n_keys * 2 sqrt(max |z|^2): capped at 40 for ordinary
    data, lower when many keys or large norms would overflow 2^62 - and w barely notices (resolution 2^-shift)."""
    import acnative
    Zs, _ = clustered_embeddings(4, 50, 64, 2, salt=93)
    big = (Zs * 2.0e3).astype(np.float32)                  # |z| ~ 2e3: inside fp16 range (hl32 needs |z| < 65504)
    Z = torch.from_numpy(big).to(dev)
    msq = acnative.max_sq([acnative.pack_images(Z)], dev)
    assert int(acnative.fixed_shift(msq, 4).item()) == 40
    assert int(acnative.fixed_shift(msq, 10 ** 9).item()) < 20
    w = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS)
    np.testing.assert_allclose(w.cpu().numpy(), _truth_w(big, big, 0), rtol=2e-6)
    minbuf = acnative.pair_min_sq_part(Z)
    assert torch.equal(acnative.pair_reduce(minbuf, 50, acnative.REDUCE_MEAN_OTHERS, 0, max_sq_t=msq), w)
    coarse = acnative.pair_reduce(minbuf, 50, acnative.REDUCE_MEAN_OTHERS, 0, max_sq_t=msq, n_keys=10 ** 9)
    np.testing.assert_allclose(coarse.cpu().numpy(), w.cpu().numpy(), rtol=1e-6)


def test_query_chunking_with_a_small_minbuf_budget(dev):
    """N large enough that minbuf must be chunked over the queries (config 5's code path), forced
    here with a tiny budget: chunked == unchunked."""
    import acnative
    Zs, _ = clustered_embeddings(6, 100, 128, 3, salt=56)
    Z = torch.from_numpy(Zs).to(dev)
    ref = acnative.pair_min_dist_staged(Z, Z.clone(), acnative.REDUCE_MEAN_OTHERS)
    small = acnative.pair_min_dist_staged(Z, Z.clone(), acnative.REDUCE_MEAN_OTHERS, minbuf_budget=6 * 100 * 4 * 2)
    assert torch.equal(ref, small)
    sup = acnative.pair_min_dist_staged(Z[:4], Z[2:].contiguous(), acnative.REDUCE_MIN_ALL, minbuf_budget=4 * 100 * 4)
    assert torch.equal(sup, acnative.pair_min_dist(Z[:4].contiguous(), Z[2:].contiguous(), acnative.REDUCE_MIN_ALL))


def test_ragged_edges_vs_oracle(dev):
    """P not a multiple of anything, nq != nk, D = 32 (one K step), single-image banks."""
    import acnative
    from oracle import alpha as oa
    for (nq, nk, P, D) in [(1, 1, 5, 32), (2, 5, 67, 96), (3, 2, 130, 64)]:
        Zq = torch.from_numpy(gaussish_tensor((nq, P, D), salt=60 + P, scale=1.5))
        Zk = torch.from_numpy(gaussish_tensor((nk, P, D), salt=61 + P, scale=1.5))
        ref = oa.weights_supervised(Zq, Zk).numpy()
        got = acnative.pair_min_dist(Zq.to(dev), Zk.to(dev), acnative.REDUCE_MIN_ALL).cpu().numpy()
        assert _w_close(got, ref), (nq, nk, P, D, np.abs(got - ref).max())


def test_weighted_avg_and_pdist_vs_oracle(dev):
    import acnative
    from oracle import alpha as oa, cluster as oc
    Zs, _ = clustered_embeddings(9, 196, 512, 3, salt=71)
    Z = torch.from_numpy(Zs)
    a64 = oa.matrix_alpha_unsupervised(1.0, 1, Z)
    a_ref, X_ref = oa.weighted_average(a64, Z)
    X = acnative.weighted_avg(a_ref.to(dev), Z.to(dev))
    np.testing.assert_allclose(X.cpu().numpy(), X_ref, rtol=0, atol=2e-5)
    d = acnative.pdist(torch.from_numpy(X_ref).to(dev)).cpu().numpy()
    np.testing.assert_allclose(d, oc.pdist(X_ref), rtol=1e-12, atol=0)
    # N that is not a multiple of the 32x32 tile, D not a multiple of 64
    Xr = gaussish_tensor((45, 100), salt=72)
    d = acnative.pdist(torch.from_numpy(Xr).to(dev)).cpu().numpy()
    np.testing.assert_allclose(d, oc.pdist(Xr), rtol=1e-12, atol=0)


def test_pdist_both_kernels(dev):
    """N <= 1024 takes the wave-per-2x2-pairs kernel, larger N the LDS-tiled one; odd N, D not a multiple of 64."""
    import acnative
    from oracle import cluster as oc
    for n, d in ((2, 8), (83, 4096), (167, 100), (1100, 72)):
        Xr = gaussish_tensor((n, d), salt=300 + n)
        got = acnative.pdist(torch.from_numpy(Xr).to(dev)).cpu().numpy()
        np.testing.assert_allclose(got, oc.pdist(Xr), rtol=1e-12, atol=0, err_msg=str((n, d)))
    again = acnative.pdist(torch.from_numpy(Xr).to(dev)).cpu().numpy()
    assert np.array_equal(got, again)                 # fixed summation order: bitwise reproducible


def test_errors_are_loud(dev):
    import acnative
    Z = torch.zeros(2, 8, 48, device=dev)      # D not a multiple of 32
    with pytest.raises(RuntimeError, match="multiple of 32"):
        acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acnative.pair_min_dist(torch.zeros(2, 8, 64), torch.zeros(2, 8, 64), 0)


def test_degenerate_category_sizes(dev):
    """N = 1: the unsupervised mean over "the other images" is empty - NaN, as torch.mean of an empty tensor
    in the reference (utils.py:227); N = 2: each image's only key image is the other one."""
    import acnative
    from oracle import alpha as oa
    Zs, _ = clustered_embeddings(2, 50, 64, 2, salt=91)
    Z = torch.from_numpy(Zs).to(dev)
    w1 = acnative.pair_min_dist(Z[:1].contiguous(), Z[:1].contiguous(), acnative.REDUCE_MEAN_OTHERS)
    assert torch.isnan(w1).all()          # the reference itself fails here (mean over dim 1 of an empty 1-D tensor)
    with pytest.raises((ValueError, IndexError, RuntimeError)):
        oa.weights_unsupervised(torch.from_numpy(Zs[:1]))
    w2 = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS).cpu().numpy()
    assert _w_close(w2, oa.weights_unsupervised(torch.from_numpy(Zs)).numpy())
    ws = acnative.pair_min_dist(Z[:1].contiguous(), Z[1:].contiguous(), acnative.REDUCE_MIN_ALL).cpu().numpy()
    assert _w_close(ws, oa.weights_supervised(torch.from_numpy(Zs[:1]), torch.from_numpy(Zs[1:])).numpy())
