"""One category end to end on one or several MI355X: extract -> aggregate -> alpha -> weighted
average -> (all-gather) -> pairwise distance.  This is the layer the reference does not have
(SURVEY.md 2.2): the reference runs ``make_category_data`` (examples/main.py:183-311) on a single
device with batch size 1; here the images of a category are sharded over the ranks of a
``torch.distributed`` process group (backend "nccl" = RCCL over xGMI on ROCm).

Per rank r of G (contiguous image block [lo_r, hi_r)):
    Z_r   = embed(images[lo_r:hi_r])                      no communication
    Z     = all_gather(Z_r)                               RCCL all-gather of the patch embeddings
    w_r   = pair_min_dist(Z_r, Z, self_offset=lo_r)       local queries x all keys (MFMA)
    a_r   = softmax(w_r / tau)  (fp64);  X_r = a_r . Z_r  no communication
    X, a  = all_gather(X_r), all_gather(a_r)              the all-gather north_star names
    d     = pdist(X)                                      rank 0
Supervised mode replaces the key set by the (sharded, all-gathered) normal-image bank.

`ops` is the compute seam: by default the HIP library (acnative + AnomalyClusteringCore); the
world-size-2 gloo tests substitute a CPU stand-in to exercise the sharding and collectives.
"""
import time

import numpy as np
import torch
import torch.distributed as dist


SYMMETRIC_MINBUF_BYTES = 1 << 30     # largest [N, N*P] fp32 matrix the partitioned symmetric path may allocate


def shard_bounds(n, world):
    """Contiguous blocks of ceil(n/world) images: [(lo, hi)] per rank (trailing ranks may be empty)."""
    per = (n + world - 1) // world if world > 0 else n
    return [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]


class HipOps:
    """Compute seam backed by libac_hip.so.  Every method raises if the library or the GPU is missing."""

    def __init__(self, core, batch_size=32):
        import acnative
        self.acnative = acnative
        self.core = core
        self.batch_size = batch_size
        self._zshape = None

    def embed(self, images):
        if images.shape[0] == 0:                     # empty shard (fewer images than ranks)
            if self._zshape is None:
                probe = torch.ones((1,) + tuple(self.core.input_shape), device=self.core.device)
                self._zshape = tuple(self.core.embed_device(probe, batch_size=1).shape[1:])
            return torch.empty((0,) + self._zshape, dtype=torch.float32, device=self.core.device)
        Z = self.core.embed_device(images, batch_size=self.batch_size)
        self._zshape = tuple(Z.shape[1:])
        return Z

    def pair_min_dist(self, Zq, Zk, reduce, self_offset):
        if Zq.shape[0] == 0:
            return torch.empty((0, Zq.shape[1]), dtype=torch.float32, device=Zq.device)
        return self.acnative.pair_min_dist_staged(Zq, Zk, reduce, self_offset)

    def pair_min_sq_part(self, Z, part, nparts):
        return self.acnative.pair_min_sq_part(Z, part, nparts)

    def pair_reduce(self, minbuf, P, reduce, self_first):
        return self.acnative.pair_reduce(minbuf, P, reduce, self_first)

    def alpha_softmax(self, w, tau, k):
        return self.acnative.alpha_softmax(w, tau, k)

    def weighted_avg(self, alpha_f32, Z):
        return self.acnative.weighted_avg(alpha_f32, Z)

    def pdist(self, X):
        return self.acnative.pdist(X)

    def sync(self):
        torch.cuda.synchronize()


def _world(group):
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def all_gather_rows(local, counts, group=None):
    """Concatenate per-rank row blocks of unequal length (counts[r] rows on rank r) on every rank.
    Pads to the largest block so one all_gather_into_tensor (one RCCL all-gather) moves everything."""
    rank, world = _world(group)
    if world == 1:
        return local
    cap = max(counts)
    tail = local.shape[1:]
    send = local
    if local.shape[0] < cap:
        send = torch.zeros((cap,) + tuple(tail), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * cap,) + tuple(tail), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(c == cap for c in counts):
        return recv
    return torch.cat([recv[r * cap: r * cap + counts[r]] for r in range(world)], dim=0)


def select_bank(n_train, n_test, train_ratio, seed=0):
    """Indices of the normal-image bank.  The reference embeds the shuffled TRAIN loader and keeps the
    first int(train_ratio * N_test) images (examples/main.py:274-281: `len(Z)` is the TEST count, and the
    shuffle is unseeded); here the shuffle is a seeded permutation, so runs are reproducible."""
    m = min(n_train, int(train_ratio * n_test))
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(n_train, generator=g)[:m]


def run_category(ops, images, supervised="unsupervised", tau=1.0, k=1, train_images=None, train_ratio=1.0,
                 group=None, bank_seed=0, want_pdist=True, want_host=True, timings=None):
    """images: [N, 3, H, W] (every rank passes the same N; only the rank's shard is read).
    `tau` may be a list: the embeddings and the patch distances are computed once and every tau gets its
    own (alpha, X) in out["by_tau"][tau]; the top-level keys are those of the first tau.
    Returns dict(alpha fp32 [N, 1, P] on device, X float32 ndarray [N, D], X_dev, pdist (rank 0), bounds)."""
    rank, world = _world(group)
    n = images.shape[0]
    bounds = shard_bounds(n, world)
    lo, hi = bounds[rank]
    counts = [b - a for a, b in bounds]
    t = {} if timings is None else timings

    def mark(name, t0):
        if timings is not None:
            ops.sync()
            t[name] = t.get(name, 0.0) + (time.perf_counter() - t0)

    t0 = time.perf_counter()
    Z_loc = ops.embed(images[lo:hi])                                   # [n_loc, P, D]
    mark("embed", t0)

    t0 = time.perf_counter()
    if supervised == "supervised":
        if train_images is None:
            raise ValueError("supervised alpha needs train_images")
        sel = select_bank(train_images.shape[0], n, train_ratio, bank_seed)
        bb = shard_bounds(len(sel), world)
        blo, bhi = bb[rank]
        bank_loc = ops.embed(train_images[sel[blo:bhi]])
        bank = all_gather_rows(bank_loc, [b - a for a, b in bb], group)
        w_loc = ops.pair_min_dist(Z_loc, bank, 1, 0)
    elif supervised == "unsupervised":
        Z_all = all_gather_rows(Z_loc, counts, group)
        P = Z_all.shape[1]
        if world > 1 and n * n * P * 4 <= SYMMETRIC_MINBUF_BYTES:
            # every rank takes 1/world of the UPPER-TRIANGLE tiles of the symmetric N*P x N*P problem
            # (each tile reduced in both directions), then one MIN all-reduce of the [N, N*P] matrix of
            # per-(key image, query patch) minima completes it: half the MFMA work of "my queries x all keys"
            minbuf = ops.pair_min_sq_part(Z_all, rank, world)
            dist.all_reduce(minbuf, op=dist.ReduceOp.MIN, group=group)
            w_loc = ops.pair_reduce(minbuf[:, lo * P: hi * P].contiguous(), P, 0, lo)
        else:
            w_loc = ops.pair_min_dist(Z_loc, Z_all, 0, lo)
    else:                                                              # "average" (main.py:290-291)
        w_loc = None
    mark("pair_min_dist", t0)

    # w does not depend on tau (and Z not on the mode): a tau sweep - the reference re-runs everything per
    # tau (examples/main.py:346-366, README "tau_list") - only repeats the cheap tail below.
    taus = list(tau) if isinstance(tau, (list, tuple)) else [tau]
    by_tau = {}
    for tv in taus:
        t0 = time.perf_counter()
        if w_loc is None:
            alpha_loc = torch.ones(Z_loc.shape[0], Z_loc.shape[1], device=Z_loc.device) / Z_loc.shape[1]
        else:
            alpha_loc = ops.alpha_softmax(w_loc, tv, k)
        alpha_f32 = alpha_loc.float().contiguous()                     # main.py:294 `.float()`
        X_loc = ops.weighted_avg(alpha_f32, Z_loc)
        X = all_gather_rows(X_loc, counts, group)
        alpha = all_gather_rows(alpha_f32, counts, group).unsqueeze(1)
        mark("alpha_softmax_weighted_avg_gather", t0)
        res = {"alpha": alpha, "X_dev": X}
        t0 = time.perf_counter()
        if want_pdist and rank == 0:
            res["pdist"] = ops.pdist(X)
        mark("pdist", t0)
        if want_host:
            res["X"] = np.array(X.cpu().numpy(), dtype=np.float32, copy=True)  # main.py:294-296 `.cpu()` + np.array
        by_tau[float(tv)] = res
    out = dict(by_tau[float(taus[0])])
    out["bounds"] = bounds
    out["by_tau"] = by_tau
    return out
