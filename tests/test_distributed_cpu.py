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


class ToyOps:
    """CPU stand-in for HipOps (tests only): tiny embedding + the oracle's alpha arithmetic."""
    P, D = 12, 32

    def embed(self, images):
        n = images.shape[0]
        if n == 0:
            return torch.zeros(0, self.P, self.D)
        flat = images.reshape(n, -1)[:, : self.P * self.D].reshape(n, self.P, self.D)
        return (flat * 1.5 + flat.roll(1, dims=2) * 0.5).contiguous()

    def pair_min_dist(self, Zq, Zk, reduce, self_offset):
        out = []
        for i in range(Zq.shape[0]):
            cols = [(torch.cdist(Zq[i], Zk[j]).min(1)[0] ** 2).sqrt() for j in range(Zk.shape[0])
                    if not (reduce == 0 and j == i + self_offset)]
            cols = torch.stack(cols, 1)
            out.append(cols.mean(1) if reduce == 0 else cols.min(1)[0])
        return torch.stack(out) if out else torch.zeros(0, Zq.shape[1])

    def pair_min_sq_part(self, Z, part, nparts):
        """Slice = the key images j with j % nparts == part (any disjoint cover works for the MIN all-reduce)."""
        n, P, _ = Z.shape
        out = torch.full((n, n * P), float("inf"))
        for j in range(part, n, nparts):
            for i in range(n):
                out[j, i * P:(i + 1) * P] = torch.cdist(Z[i], Z[j]).min(1)[0] ** 2
        return out

    def pair_reduce(self, minbuf, P, reduce, self_first):
        nk, cols = minbuf.shape
        d = minbuf.sqrt().reshape(nk, cols // P, P)
        out = []
        for i in range(cols // P):
            rows = [d[j, i] for j in range(nk) if not (reduce == 0 and j == i + self_first)]
            rows = torch.stack(rows, 1)
            out.append(rows.mean(1) if reduce == 0 else rows.min(1)[0])
        return torch.stack(out)

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


def _inputs(n=7, n_train=9):
    g = torch.Generator().manual_seed(11)
    return torch.randn(n, 3, 16, 16, generator=g), torch.randn(n_train, 3, 16, 16, generator=g)


def _worker(rank, world, init_file, mode, out_dir):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import run_category
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    images, train = _inputs()
    out = run_category(ToyOps(), images, mode, tau=1.0, train_images=train, train_ratio=1.0)
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


def test_all_gather_rows_single_process_passthrough():
    from patchcore.pipeline import all_gather_rows
    t = torch.arange(6.0).reshape(3, 2)
    assert all_gather_rows(t, [3]) is t
