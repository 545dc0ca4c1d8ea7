"""CPU, world_size 2 over gloo: the sharding + all-gather layer of patchcore.pipeline.

The compute seam is replaced by a CPU stand-in (oracle arithmetic on a toy embedding), so what is
tested is the distributed logic: shard bounds, padded all-gather of unequal shards, self_offset of
the query shard, seeded bank selection - a 2-rank run must reproduce the 1-rank run exactly."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, MODELS


INF_BITS = 0x7f800000


class ToyPacked:
    """CPU stand-in for acnative.Packed: the 'packed' image is Z itself, rows image-major."""

    def __init__(self, hl, sq, n, P, D):
        self.hl, self.sq, self.n, self.P, self.D = hl, sq, n, P, D

    def slice(self, lo, hi):
        P = self.P
        return ToyPacked(self.hl[lo * P: hi * P], self.sq[lo * P: hi * P], hi - lo, P, self.D)

    def tensors(self):
        return [self.hl, self.sq]


class ToyOps:
    """CPU stand-in for HipOps (tests only): tiny embedding + the oracle's alpha arithmetic behind the same
    block primitives (pack / pair_block / pair_accumulate in 64-bit fixed point / pair_finalize)."""
    P, D = 12, 32

    def embed(self, images):
        n = images.shape[0]
        if n == 0:
            return torch.zeros(0, self.P, self.D)
        flat = images.reshape(n, -1)[:, : self.P * self.D].reshape(n, self.P, self.D)
        return (flat * 1.5 + flat.roll(1, dims=2) * 0.5).contiguous()

    def pack(self, Z):
        n, P, D = Z.shape
        flat = Z.reshape(n * P, D).contiguous()
        return ToyPacked(flat, (flat.double() ** 2).sum(1).float(), n, P, D)

    def empty_packed(self, n, like):
        return ToyPacked(torch.empty(n * like.P, like.D), torch.empty(n * like.P), n, like.P, like.D)

    def max_sq(self, packed_list, device, out=None):
        m = torch.zeros(1) if out is None else out.zero_()
        for pk in packed_list:
            if pk.n:
                m.copy_(torch.maximum(m, pk.sq.max().reshape(1)))
        return m

    def fixed_shift(self, max_sq_t, n_keys):
        bound = max(int(n_keys), 1) * 2.0 * float(max_sq_t.double().sqrt()) + 1.0
        e = np.frexp(bound)[1]
        return torch.tensor([max(-60, min(40, 62 - int(e)))], dtype=torch.int32)

    def new_acc(self, ncols, reduce, device):
        if reduce == 0:
            return torch.zeros(ncols, dtype=torch.int64)
        return torch.full((ncols,), INF_BITS, dtype=torch.int32)

    launches = 0          # pair_block calls of this process (the eight-rank test counts them)

    def pair_block(self, keys, queries, two_way=False, part=0, nparts=1):
        ToyOps.launches += 1
        P = keys.P
        same = keys.hl.data_ptr() == queries.hl.data_ptr() and keys.n == queries.n
        Zk, Zq = keys.hl.reshape(keys.n, P, -1), queries.hl.reshape(queries.n, P, -1)
        mb = torch.full((keys.n, queries.n * P), float("inf"))
        mbt = torch.full((queries.n, keys.n * P), float("inf")) if two_way else None
        for j in range(keys.n):
            if same and j % nparts != part:            # any disjoint cover works for the MIN all-reduce
                continue
            for i in range(queries.n):
                d2 = torch.cdist(Zq[i], Zk[j]) ** 2          # [P queries, P keys]
                mb[j, i * P:(i + 1) * P] = d2.min(1)[0]
                if two_way:
                    mbt[i, j * P:(j + 1) * P] = d2.min(0)[0]
        mb = mb.view(torch.int32)
        return (mb, mbt.view(torch.int32)) if two_way else mb

    def pair_accumulate(self, minbuf, P, reduce, self_first, shift, acc):
        nk, ncols = minbuf.shape
        if nk == 0 or ncols == 0:
            return
        if reduce == 0:
            v = torch.round(minbuf.view(torch.float32).sqrt().double() * (2.0 ** int(shift))).to(torch.int64)
            j = torch.arange(nk).reshape(nk, 1)
            self_of_col = (self_first + torch.arange(ncols) // P).reshape(1, ncols)
            acc += torch.where(j == self_of_col, torch.zeros_like(v), v).sum(0)
        else:
            torch.minimum(acc, minbuf.min(0)[0], out=acc)

    def pair_finalize(self, acc, P, reduce, count, shift):
        if reduce == 0:
            w = (acc.double() * (2.0 ** -int(shift)) / count).float() if count else torch.full(acc.shape, float("nan"))
        else:
            w = acc.view(torch.float32).sqrt()
        return w.reshape(-1, P)

    def alpha_softmax(self, w, tau, k):
        from oracle import alpha as oa
        if w.shape[0] == 0:
            return torch.zeros(0, w.shape[1], dtype=torch.float64)
        return torch.cat([oa.alpha_from_weights(wi, tau, k) for wi in w], 0)

    def weighted_avg(self, alpha_f32, Z):
        return torch.bmm(alpha_f32.unsqueeze(1), Z).squeeze(1)

    def pdist(self, X):
        from oracle import cluster as oc
        return torch.from_numpy(oc.pdist(X.numpy()))

    def sync(self):
        pass


class ToyPrunedOps(ToyOps):
    """ToyOps + a CPU stand-in for the pruned nearest-neighbour search of the supervised mode (pipeline.
    alpha_weights_supervised): the "first pass" is exact here, but every `overflow_every`-th query row is declared
    overflowed on every bank shard - on a subset of the ranks only when `ranks` is given - so that the ONE-flag decision,
    its MAX all-reduce and the second trip of the bank shards round the ring all run under gloo."""
    nn_prune = True

    def __init__(self, overflow_every=0, ranks=None):
        self.overflow_every, self.ranks = overflow_every, ranks

    def h16(self, pk):
        return pk.hl

    def nn_slack(self, mk, mq):
        return torch.zeros(3)

    def nn_state(self, nq_rows, device, overflow=None):
        class State:
            pass
        st = State()
        st.n = nq_rows
        st.best = torch.full((nq_rows,), float("inf"), dtype=torch.float64)
        st.overflow = torch.zeros(1, dtype=torch.int32) if overflow is None else overflow.zero_()
        st.overflow_rows = torch.zeros(nq_rows, dtype=torch.int32)
        return st

    def _forced(self, n):
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else 0
        if not self.overflow_every or (self.ranks is not None and rank not in self.ranks):
            return torch.zeros(n, dtype=torch.bool)
        return (torch.arange(n) % self.overflow_every) == 0

    def nn_scan(self, state, keys, keys_h16, queries, queries_h16, slack):
        d2 = (torch.cdist(queries.hl, keys.hl) ** 2).min(1)[0].double()
        over = self._forced(state.n)
        state.best = torch.where(over, state.best, torch.minimum(state.best, d2))
        if over.any():
            state.overflow[0] = 1
            state.overflow_rows[over] = 1

    def nn_any_overflow(self, state):
        return bool(int(state.overflow.item()))

    def nn_overflow_queries(self, state, queries):
        ops = self

        class Over:
            def __init__(self):
                self.idx = state.overflow_rows.nonzero().flatten()
                self.best = torch.full((self.idx.numel(),), float("inf"), dtype=torch.float64)

            def scan(self, keys, budget=None):
                if self.idx.numel() and keys is not None and keys.n:
                    d2 = (torch.cdist(queries.hl[self.idx], keys.hl) ** 2).min(1)[0].double()
                    self.best = torch.minimum(self.best, d2)

            def commit(self, st):
                st.best[self.idx] = self.best
        return Over()

    def nn_finalize(self, state, P):
        return state.best.sqrt().float().reshape(-1, P)


def _worker_pruned(rank, world, init_file, out_dir, overflow_every, ranks):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import run_category
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    images, train = _inputs()
    out = run_category(ToyPrunedOps(overflow_every, ranks), images, "supervised", tau=1.0, train_images=train, train_ratio=1.0)
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overflow_every,ranks", [(3, 0, None), (3, 5, None), (4, 3, [2])])
def test_supervised_pruned_search_collectives(world, overflow_every, ranks):
    """The pruned search's distributed logic on a CPU stand-in: no overflow (one ring pass, one flag read), overflow on
    every rank, and overflow on ONE rank only - the others must still join the second trip of the shards round the ring
    (the flag is MAX all-reduced).  Every case equals the un-pruned one-rank result."""
    from patchcore.pipeline import run_category
    images, train = _inputs()
    ref = run_category(ToyOps(), images, "supervised", tau=1.0, train_images=train, train_ratio=1.0)
    one = run_category(ToyPrunedOps(overflow_every), images, "supervised", tau=1.0, train_images=train, train_ratio=1.0)
    np.testing.assert_allclose(one["alpha"].numpy(), ref["alpha"].numpy(), rtol=1e-6)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_pruned, args=(world, os.path.join(d, "rdzv"), d, overflow_every, ranks), nprocs=world, join=True)
        for r in range(world):
            np.testing.assert_allclose(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].numpy(), rtol=1e-6)


def _inputs(n=7, n_train=9):
    g = torch.Generator().manual_seed(11)
    return torch.randn(n, 3, 16, 16, generator=g), torch.randn(n_train, 3, 16, 16, generator=g)


def _worker(rank, world, init_file, mode, out_dir, schedule="auto", budget=1 << 30, remote_bytes=8 << 30):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import run_category
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    images, train = _inputs()
    out = run_category(ToyOps(), images, mode, tau=1.0, train_images=train, train_ratio=1.0, schedule=schedule,
                       budget=budget, remote_bytes=remote_bytes)
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].numpy())
    np.save(os.path.join(out_dir, "X_%d.npy" % rank), out["X"])
    if rank == 0:
        np.save(os.path.join(out_dir, "pdist.npy"), out["pdist"].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["unsupervised", "supervised", "average"])
def test_two_ranks_equal_one_rank(mode):
    from patchcore.pipeline import run_category
    images, train = _inputs()
    ref = run_category(ToyOps(), images, mode, tau=1.0, train_images=train, train_ratio=1.0)
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mp.spawn(_worker, args=(2, init_file, mode, d), nprocs=2, join=True)
        for r in range(2):
            np.testing.assert_array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].numpy())
            np.testing.assert_array_equal(np.load(os.path.join(d, "X_%d.npy" % r)), ref["X"])
        np.testing.assert_array_equal(np.load(os.path.join(d, "pdist.npy")), ref["pdist"].numpy())
    assert ref["alpha"].shape == (7, 1, ToyOps.P) and ref["X"].shape == (7, ToyOps.D)
    np.testing.assert_allclose(ref["alpha"].sum(-1).numpy(), 1.0, rtol=1e-6)


@pytest.mark.parametrize("world,schedule,budget", [(2, "ring", 1 << 30), (2, "ring", 2 * 2 * 12 * 4), (3, "ring", 12 * 4),
                                                   (3, "slices", 1 << 30), (4, "ring", 3 * 3 * 12 * 4)])
def test_unsupervised_schedules_are_bit_identical(world, schedule, budget):
    """The ring schedule (whole shards, or forced down to 1-3 image sub-blocks by a tiny budget so that the
    step-G/2 split, the two-way blocks and the int64 all-reduce all engage) and the slices schedule give the
    SAME BITS as one rank with one block: the fixed-point sum makes the order irrelevant.  The oracle's fp32
    mean is within 1e-6 of it."""
    from patchcore.pipeline import run_category
    from oracle import alpha as oa
    images, train = _inputs()
    ref = run_category(ToyOps(), images, "unsupervised", tau=1.0)
    one_rank_blocked = run_category(ToyOps(), images, "unsupervised", tau=1.0, budget=2 * 2 * 12 * 4)
    assert torch.equal(one_rank_blocked["w_loc"], ref["w_loc"])
    w_oracle = oa.weights_unsupervised(ToyOps().embed(images))
    np.testing.assert_allclose(ref["w_loc"].numpy(), w_oracle.numpy(), rtol=2e-6, atol=2e-6)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, os.path.join(d, "rdzv"), "unsupervised", d, schedule, budget), nprocs=world,
                 join=True)
        for r in range(world):
            np.testing.assert_array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].numpy())
            np.testing.assert_array_equal(np.load(os.path.join(d, "X_%d.npy" % r)), ref["X"])


def test_supervised_ring_with_sub_blocks():
    from patchcore.pipeline import run_category
    images, train = _inputs()
    ref = run_category(ToyOps(), images, "supervised", tau=1.0, train_images=train, train_ratio=1.0)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(3, os.path.join(d, "rdzv"), "supervised", d, "auto", 2 * 2 * 12 * 4), nprocs=3, join=True)
        for r in range(3):
            np.testing.assert_array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].numpy())


def test_ring_with_two_receive_buffers():
    """remote_bytes = 1 forces the double-buffered ring (what N = 10 000 runs): with 4 ranks the bank shards of the
    supervised mode need 3 steps through 2 buffers, so a buffer is refilled while the run is in flight."""
    from patchcore.pipeline import run_category
    images, train = _inputs()
    ref = run_category(ToyOps(), images, "supervised", tau=1.0, train_images=train, train_ratio=1.0)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(4, os.path.join(d, "rdzv"), "supervised", d, "auto", 1 << 30, 1), nprocs=4, join=True)
        for r in range(4):
            np.testing.assert_array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].numpy())


def test_all_gather_rows_single_process_passthrough():
    from patchcore.pipeline import all_gather_rows
    t = torch.arange(6.0).reshape(3, 2)
    assert all_gather_rows(t, [3]) is t


# ------------------------------------------------------------------ several categories as one job (run_categories)
def _category_items(sizes=(7, 3, 5, 1), n_train=6):
    g = torch.Generator().manual_seed(23)
    return [{"images": torch.randn(n, 3, 16, 16, generator=g), "train_images": torch.randn(n_train + c, 3, 16, 16, generator=g)}
            for c, n in enumerate(sizes)]


def _worker_categories(rank, world, init_file, out_dir, mode, schedule, ops_kind, kw):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import run_categories
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    ops = ToyPrunedOps(*ops_kind) if ops_kind is not None else ToyOps()
    outs = run_categories(ops, _category_items(), mode, tau=[1.0, 0.5], train_ratio=1.0, schedule=schedule, **kw)
    for c, o in enumerate(outs):
        for tv, res in o["by_tau"].items():
            np.save(os.path.join(out_dir, "alpha_%d_%d_%s.npy" % (rank, c, tv)), res["alpha"].numpy())
            np.save(os.path.join(out_dir, "X_%d_%d_%s.npy" % (rank, c, tv)), res["X"])
            if rank == 0:
                np.save(os.path.join(out_dir, "pdist_%d_%s.npy" % (c, tv)), res["pdist"].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,schedule,ops_kind,kw", [
    (2, "unsupervised", "pipeline", None, {}),
    (3, "unsupervised", "pipeline", None, {"budget": 2 * 2 * 12 * 4}),              # sub-blocked steps, odd world, empty shards
    (4, "unsupervised", "pipeline", None, {"group_images": 3}),                       # several groups of categories
    (3, "unsupervised", "pipeline", None, {"ring_bytes_all": 1}),                     # rings too large to be in flight together
    (2, "supervised", "pipeline", None, {}),                                          # ToyOps has no pruned search: exhaustive blocks
    (3, "supervised", "pipeline", (0, None), {}),                                     # pruned search, no overflow
    (3, "supervised", "pipeline", (5, [1]), {}),                                      # overflow on ONE rank: every rank joins the second trip
    (2, "average", "pipeline", None, {}),
    (3, "unsupervised", "per_gpu", None, {}),
    (5, "unsupervised", "per_gpu", None, {}),                                         # more ranks than categories: an idle rank
    (2, "supervised", "per_gpu", (0, None), {}),
])
def test_categories_as_one_job_equal_the_per_category_loop(world, mode, schedule, ops_kind, kw):
    """run_categories - the cross-category pipeline (joint ViT batches, every ring posted up front, step 0 of all
    categories first, one MAX / SUM / flag all-reduce per group) and the category-per-GPU schedule - gives, on every
    rank and for every tau, the SAME BITS as run_category called per category on one rank (the reference's loop,
    examples/main.py:346-366)."""
    from patchcore.pipeline import run_category, run_categories
    items = _category_items()
    ref = [run_category(ToyOps(), it["images"], mode, tau=[1.0, 0.5], train_images=it["train_images"], train_ratio=1.0)
           for it in items]
    one = run_categories(ToyPrunedOps(*ops_kind) if ops_kind is not None else ToyOps(), items, mode, tau=[1.0, 0.5],
                         train_ratio=1.0, schedule="pipeline", **kw)
    exact = ops_kind is None                      # the toy pruned search takes its minimum in another order of operations
    for o, r in zip(one, ref):
        for tv in r["by_tau"]:
            if exact:
                np.testing.assert_array_equal(o["by_tau"][tv]["alpha"].numpy(), r["by_tau"][tv]["alpha"].numpy())
            else:
                np.testing.assert_allclose(o["by_tau"][tv]["alpha"].numpy(), r["by_tau"][tv]["alpha"].numpy(), rtol=1e-6)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_categories, args=(world, os.path.join(d, "rdzv"), d, mode, schedule, ops_kind, kw), nprocs=world, join=True)
        for rk in range(world):
            for c, r in enumerate(ref):
                for tv, res in r["by_tau"].items():
                    a = np.load(os.path.join(d, "alpha_%d_%d_%s.npy" % (rk, c, tv)))
                    X = np.load(os.path.join(d, "X_%d_%d_%s.npy" % (rk, c, tv)))
                    if exact:
                        np.testing.assert_array_equal(a, res["alpha"].numpy())
                        np.testing.assert_array_equal(X, res["X"])
                    else:
                        np.testing.assert_allclose(a, res["alpha"].numpy(), rtol=1e-6)
                        np.testing.assert_allclose(X, res["X"], rtol=1e-5, atol=1e-6)
        if exact:
            for c, r in enumerate(ref):
                for tv, res in r["by_tau"].items():
                    np.testing.assert_array_equal(np.load(os.path.join(d, "pdist_%d_%s.npy" % (c, tv))), res["pdist"].numpy())


# ------------------------------------------------------------------ the ring carries taps instead of packed embeddings
class ToyTapsOps(ToyOps):
    """ToyOps whose ring payload is the "tapped feature maps" (here: the flattened pixels the toy embedding reads):
    embed(..., want_taps=True) hands them out and taps_to_packed re-embeds + packs a received block - the seam of
    HipOps.taps_to_packed (aggregate + pack), so the taps ring, its receive buffers and its decode step run under gloo."""
    ring_payload = "taps"

    def embed(self, images, want_taps=False):
        from patchcore.pipeline import Taps
        Z = ToyOps.embed(self, images)
        if not want_taps:
            return Z
        n = images.shape[0]
        flat = images.reshape(n, -1)[:, : self.P * self.D].contiguous() if n else torch.zeros(0, self.P * self.D)
        return Z, Taps([flat], n)

    def _z(self, taps):
        f = taps.feats[0].reshape(taps.n, self.P, self.D)
        return (f * 1.5 + f.roll(1, dims=2) * 0.5).contiguous()

    def taps_to_packed(self, taps):
        return self.pack(self._z(taps))

    # the seam of run_category's `lean` mode (HipOps.embed_taps / pack_taps / weighted_avg_taps): Z is never kept
    def zshape(self):
        return (self.P, self.D)

    def embed_taps(self, images):
        return self.embed(images, want_taps=True)[1]

    pack_taps = taps_to_packed

    def weighted_avg_taps(self, alpha_f32, taps):
        return self.weighted_avg(alpha_f32.reshape(taps.n, self.P), self._z(taps))    # (n = 0: a rank without images)


class ToyTapsPrunedOps(ToyTapsOps, ToyPrunedOps):
    pass


def _worker_taps(rank, world, init_file, out_dir, mode, many, kw):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import run_category, run_categories
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    ops = ToyTapsPrunedOps(0, None) if mode == "supervised_pruned" else ToyTapsOps()
    m = "supervised" if mode.startswith("supervised") else mode
    kw = dict(kw)
    lean_rank, again = kw.pop("_lean_rank", None), kw.pop("_again_with_taps", False)
    if lean_rank is not None:                      # ONE rank is short of memory: every rank must leave the one-job order
        ops.wants_lean = lambda _n: rank == lean_rank
    if many:
        outs = run_categories(ops, _category_items(), m, tau=1.0, train_ratio=1.0, **kw)
        if again:                                  # a second pass over the embeddings the lean first pass handed back (Taps)
            from patchcore.pipeline import Taps
            assert any(isinstance(o["Z_loc"], Taps) for o in outs) == (rank == lean_rank)   # only that rank dropped Z
            outs = run_categories(ops, _category_items(), m, tau=1.0, train_ratio=1.0, Z_locs=[o["Z_loc"] for o in outs], **kw)
    else:
        images, train = _inputs()
        outs = [run_category(ops, images, m, tau=1.0, train_images=train, train_ratio=1.0, **kw)]
    for c, o in enumerate(outs):
        np.save(os.path.join(out_dir, "alpha_%d_%d.npy" % (rank, c)), o["alpha"].numpy())
        np.save(os.path.join(out_dir, "X_%d_%d.npy" % (rank, c)), o["X"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,many,kw", [
    (2, "unsupervised", False, {}),
    (4, "unsupervised", False, {"budget": 2 * 2 * 12 * 4}),
    (3, "supervised", False, {}),
    (4, "supervised", False, {"remote_bytes": 1}),            # double-buffered ring: a receive buffer is decoded, then refilled
    (3, "supervised_pruned", False, {}),
    (3, "unsupervised", True, {}),
    (4, "supervised_pruned", True, {}),
    (3, "unsupervised", False, {"lean": True}),               # Z dropped: packed blocks and X rebuilt from the taps
    (2, "supervised", False, {"lean": True}),
    (2, "average", False, {"lean": True}),
    (3, "unsupervised", True, {"_lean_rank": 1}),             # one-job order, one rank wants lean: all fall back together
    (2, "supervised", True, {"_lean_rank": 0, "_again_with_taps": True}),   # ... and the Taps come back in as Z_locs
])
def test_ring_of_taps_equals_ring_of_packed_shards(world, mode, many, kw):
    """Several ranks send the tapped feature maps round the ring (2.7x fewer bytes on the real shapes) and the receiver
    re-aggregates + packs them; the result is the one-rank result of the plain seam, bit for bit (the pruned toy search
    takes its minimum in another order of operations: 1e-6)."""
    from patchcore.pipeline import run_category
    m = "supervised" if mode.startswith("supervised") else mode
    if many:
        sets = [(it["images"], it["train_images"]) for it in _category_items()]
    else:
        sets = [_inputs()]
    ref = [run_category(ToyOps(), a, m, tau=1.0, train_images=b, train_ratio=1.0) for a, b in sets]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_taps, args=(world, os.path.join(d, "rdzv"), d, mode, many, kw), nprocs=world, join=True)
        for rk in range(world):
            for c, r in enumerate(ref):
                a = np.load(os.path.join(d, "alpha_%d_%d.npy" % (rk, c)))
                X = np.load(os.path.join(d, "X_%d_%d.npy" % (rk, c)))
                if mode == "supervised_pruned":
                    np.testing.assert_allclose(a, r["alpha"].numpy(), rtol=1e-6)
                    np.testing.assert_allclose(X, r["X"], rtol=1e-5, atol=1e-6)
                else:
                    np.testing.assert_array_equal(a, r["alpha"].numpy())
                    np.testing.assert_array_equal(X, r["X"])


def _worker_rot(rank, world, init_file, out_dir, mode, rot):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import run_category
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    images, train = _inputs()
    out = run_category(ToyTapsOps(), images, mode, tau=1.0, train_images=train, train_ratio=1.0, rot=rot)
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].numpy())
    np.save(os.path.join(out_dir, "X_%d.npy" % rank), out["X"])
    np.save(os.path.join(out_dir, "n_%d.npy" % rank), np.array([out["Z_loc"].shape[0]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,rot", [(3, "unsupervised", 1), (4, "unsupervised", 3), (3, "supervised", 2), (4, "average", 5)])
def test_rotated_shards_give_the_same_result(world, mode, rot):
    """run_category(rot=...): rank r works the shard of virtual rank (r + rot) % world - the ring's peers, the split step
    and the all-gathers follow - and alpha / X come out in image order, bit-equal to one rank."""
    from patchcore.pipeline import run_category, shard_bounds
    images, train = _inputs()
    ref = run_category(ToyOps(), images, mode, tau=1.0, train_images=train, train_ratio=1.0)
    bounds = shard_bounds(images.shape[0], world)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_rot, args=(world, os.path.join(d, "rdzv"), d, mode, rot), nprocs=world, join=True)
        for r in range(world):
            np.testing.assert_array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].numpy())
            np.testing.assert_array_equal(np.load(os.path.join(d, "X_%d.npy" % r)), ref["X"])
            lo, hi = bounds[(r + rot) % world]
            assert int(np.load(os.path.join(d, "n_%d.npy" % r))[0]) == hi - lo


def _worker_eight(rank, world, init_file, out_dir, mode, many):
    sys.path[:0] = [ROOT, MODELS]
    torch.set_num_threads(1)
    from patchcore.pipeline import run_category, run_categories
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    g = torch.Generator().manual_seed(31)
    images, train = torch.randn(19, 3, 16, 16, generator=g), torch.randn(21, 3, 16, 16, generator=g)
    ops = ToyTapsPrunedOps(7, [3]) if mode == "supervised" else ToyTapsOps()
    if many:
        items = [{"images": images, "train_images": train}, {"images": images[:5], "train_images": train[:9]},
                 {"images": images[3:16], "train_images": train[2:]}]
        outs = run_categories(ops, items, mode, tau=1.0, train_ratio=1.0)
    else:
        outs = [run_category(ops, images, mode, tau=1.0, train_images=train, train_ratio=1.0)]
    if rank in (0, 5, 7):
        for c, o in enumerate(outs):
            np.save(os.path.join(out_dir, "alpha_%d_%d.npy" % (rank, c)), o["alpha"].numpy())
    np.save(os.path.join(out_dir, "launches_%d.npy" % rank), np.array([ToyOps.launches]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,many,merge", [("unsupervised", False, "1"), ("unsupervised", False, "0"), ("supervised", False, "1"),
                                             ("supervised", False, "0"), ("unsupervised", True, "1")])
def test_eight_ranks_as_the_driver_will_launch_them(mode, many, merge, monkeypatch):
    """World size 8 - what `bench.py --gpus 8` runs - over gloo with the taps ring: 19 images give the shards 3 3 3 3 3 3 1 0
    (a short one and an EMPTY one), four ring steps with the split step 4 (unsupervised), seven bank-shard steps and an
    overflow raised on rank 3 only (supervised), rotated shards over three categories (many).  Equal to one rank - with the
    later ring steps fused into rectangles (round 6: pipeline.merge_later_steps; the ranks of the lower half then run TWO
    launches, their own triangle and one rectangle) and step by step (AC_RING_MERGE=0)."""
    from patchcore.pipeline import run_category
    monkeypatch.setenv("AC_RING_MERGE", merge)
    g = torch.Generator().manual_seed(31)
    images, train = torch.randn(19, 3, 16, 16, generator=g), torch.randn(21, 3, 16, 16, generator=g)
    sets = [(images, train), (images[:5], train[:9]), (images[3:16], train[2:])] if many else [(images, train)]
    ref = [run_category(ToyOps(), a, mode, tau=1.0, train_images=b, train_ratio=1.0) for a, b in sets]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_eight, args=(8, os.path.join(d, "rdzv"), d, mode, many), nprocs=8, join=True)
        if mode == "unsupervised" and not many:
            n_launch = [int(np.load(os.path.join(d, "launches_%d.npy" % r))[0]) for r in range(8)]
            # shards 3 3 3 3 3 3 1 0.  Step by step: own triangle + one block per step with a non-empty partner.  Fused: ranks 0-3
            # one rectangle; rank 4 (5, 6 | its last image x 0), 5 (6 | 0 | its last image x 1) two and three, rank 6 (one image: 0, 1) one; rank 7 is empty.
            assert n_launch == ([2, 2, 2, 2, 3, 4, 2, 0] if merge == "1" else [5, 5, 5, 4, 4, 4, 3, 0]), n_launch
        for rk in (0, 5, 7):
            for c, r in enumerate(ref):
                a = np.load(os.path.join(d, "alpha_%d_%d.npy" % (rk, c)))
                if mode == "supervised":
                    np.testing.assert_allclose(a, r["alpha"].numpy(), rtol=1e-6)
                else:
                    np.testing.assert_array_equal(a, r["alpha"].numpy())
