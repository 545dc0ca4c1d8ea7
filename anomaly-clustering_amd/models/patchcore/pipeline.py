"""One category end to end on one or several MI355X: extract -> aggregate -> alpha -> weighted
average -> (all-gather) -> pairwise distance.  This is the layer the reference does not have
(SURVEY.md 2.2): the reference runs ``make_category_data`` (examples/main.py:183-311) on a single
device with batch size 1 and walks all N(N-1) ordered image pairs one ``cdist`` at a time
(models/patchcore/utils.py:222-237); here the images of a category are sharded over the ranks of a
``torch.distributed`` process group (backend "nccl" = RCCL over xGMI on ROCm) and the pair loop
becomes a block schedule over packed (split-fp16) image blocks.

Per rank r of G (contiguous image block [lo_r, hi_r)):
    Z_r      = embed(images[lo_r:hi_r])                   no communication
    pk_r     = pack(Z_r)                                  hl32 image + squared norms, ONCE, local shard only
    w_r      = alpha_weights(...)                         the block schedule below (MFMA)
    a_r      = softmax(w_r / tau)  (fp64);  X_r = a_r . Z_r      no communication
    X, a     = all_gather(X_r), all_gather(a_r)           the all-gather north_star names
    d        = pdist(X)                                   rank 0

Unsupervised weights: w_i[p] = mean_{j != i} min_q |Z_i[p] - Z_j[q]|.  d(i,j) block = d(j,i) block
transposed, so every UNORDERED pair of image blocks is multiplied once and reduced in both directions
(half the MFMA work of "my queries x all keys").  Two schedules:

  "ring" (the default): nothing is gathered.  Rank r keeps its shard resident and receives the shard of rank
      r+k for the ring steps k = 1 .. G/2 - HALF the shards, each over its own xGMI link, all posted up front
      for an MVTec-sized category (so they travel under the local work of step 0), double-buffered for a large
      one (N = 10 000: the transfer of step k+2 runs under the MFMA work of step k); it multiplies
      (its shard) x (that shard) two-way, in sub-blocks of c x c images sized to the minbuf budget, and for even
      G the step k = G/2 block is split between its two owners.  Sums for the other rank's queries go into the same [N*P] int64 accumulator;
      ONE all-reduce(SUM) of it at the end hands every rank its complete sums.  Per-GPU bytes at N = 10 000,
      G = 8, P = 784, D = 4096: Z fp32 16 GB + hl32 16 GB + 2 x 16 GB receive buffers + 2 x 1 GiB minbuf
      + 63 MB accumulator = 66 GB of 288.

  "slices" (kept as an alternative; needs N^2 * P * 4 bytes of minima): the packed shards are all-gathered
      (one RCCL all-gather of hl32 + norms: N = 167 -> 2.1 GB on every rank), each rank runs slice r/G of the
      upper-triangle tile space of the whole N*P x N*P problem into its own [N, N*P] matrix of minima, one MIN
      all-reduce (N = 167: 87 MB) completes it.  Perfectly balanced, but it moves G - 1 shards to every rank
      before any MFMA work can start.

The sum over key images is kept in 64-bit fixed point (csrc/alpha.hip), so both schedules, any rank count
and any block size give bit-identical w for the same Z.  On one GPU the ring schedule degenerates to
"sub-blocks of my own shard" - that is how N = 2 000+ runs in a fixed memory budget.

Supervised weights (min over the normal-image bank): the bank is sharded like the images, each rank
packs its share, and the G bank shards go round the same ring.  The min over all bank patches is a
1-nearest-neighbour search, run as a pruned EXACT search: one MFMA launch per shard on the fp16 high halves
(a third of the split-fp16 work) keeps the bank patches within a proven bound of the running minimum, and
only those are measured exactly (alpha_weights_supervised; csrc/alpha.hip).  No collective at the end.

`ops` is the compute seam: by default the HIP library (acnative + AnomalyClusteringCore); the
world-size-2 gloo tests substitute a CPU stand-in to exercise the schedules and collectives.
"""
import math
import time

import numpy as np
import torch
import torch.distributed as dist


MINBUF_BYTES = 1 << 30        # largest matrix of minima one launch may fill (x2 for a two-way block)
REMOTE_BYTES = 8 << 30        # receive buffers of the ring: all steps in flight when they fit, else two buffers
MEAN, MIN = 0, 1              # acnative.REDUCE_MEAN_OTHERS / REDUCE_MIN_ALL
NO_SELF = -(1 << 62)


def shard_bounds(n, world):
    """Contiguous blocks of ceil(n/world) images: [(lo, hi)] per rank (trailing ranks may be empty)."""
    per = (n + world - 1) // world if world > 0 else n
    return [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]


def block_images(P, budget=MINBUF_BYTES):
    """Largest c with a c x c-image matrix of minima (c * c * P fp32) inside the budget."""
    return max(1, int(math.isqrt(max(1, budget // (4 * P)))))


def _subranges(lo, hi, c):
    return [(a, min(hi, a + c)) for a in range(lo, hi, c)]


def unsup_schedule(bounds, rank, c):
    """The unsupervised block schedule of one rank, in GLOBAL image indices.  Yields
         ("sym", step, (a0, a1))              keys = queries = images [a0, a1): symmetric launch
         ("two", step, (a0, a1), (b0, b1))     keys = [a0, a1), queries = [b0, b1): two-way launch
       step 0 touches only the rank's own shard; step k >= 1 needs the shard of rank (rank + k) % world,
       which holds the block that is not the rank's own.  Over all ranks every unordered pair of distinct
       images is covered exactly once (tests/test_host_logic.py checks that for many shapes)."""
    world = len(bounds)
    lo, hi = bounds[rank]
    subs = _subranges(lo, hi, c)
    for ia, a in enumerate(subs):
        yield ("sym", 0, a)
        for b in subs[ia + 1:]:
            yield ("two", 0, a, b)
    for k in range(1, world // 2 + 1):
        src = (rank + k) % world
        slo, shi = bounds[src]
        mine, theirs = (lo, hi), (slo, shi)
        if 2 * k == world:
            # the pair {rank, src} meets from both sides at this step: the lower rank takes the first half of
            # ITS images against all of the higher rank's, the higher rank the remaining images of the lower
            low, high = min(rank, src), max(rank, src)
            llo, lhi = bounds[low]
            mid = llo + (lhi - llo + 1) // 2
            if rank == low:
                mine = (llo, mid)
            else:
                theirs = (mid, lhi)
        for a in _subranges(mine[0], mine[1], c):
            for b in _subranges(theirs[0], theirs[1], c):
                yield ("two", k, a, b)


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

    # ---- packed blocks and the pair kernels (see acnative for the semantics)
    def pack(self, Z):
        return self.acnative.pack_images(Z)

    def empty_packed(self, n, like):
        return self.acnative.Packed.empty(n, like.P, like.D, like.hl.device)

    def max_sq(self, packed_list, device):
        return self.acnative.max_sq(packed_list, device)

    def fixed_shift(self, max_sq_t, n_keys):
        return self.acnative.fixed_shift(max_sq_t, n_keys)

    def new_acc(self, ncols, reduce, device):
        return self.acnative.new_acc(ncols, reduce, device)

    def pair_block(self, keys, queries, two_way=False, part=0, nparts=1):
        return self.acnative.pair_block(keys, queries, two_way, part, nparts)

    def pair_accumulate(self, minbuf, P, reduce, self_first, shift, acc):
        self.acnative.pair_accumulate(minbuf, P, reduce, self_first, shift, acc)

    def pair_finalize(self, acc, P, reduce, count, shift):
        return self.acnative.pair_finalize(acc, P, reduce, count, shift)

    # ---- pruned exact nearest neighbour of the supervised mode
    nn_prune = True
    nn_cap = None            # candidate slots per query patch and bank shard (None: acnative.NN_CAP)

    def h16(self, pk):
        return self.acnative.h16_from_packed(pk)

    def nn_slack(self, max_sq_keys, max_sq_queries):
        return self.acnative.nn_slack(max_sq_keys, max_sq_queries)

    def nn_state(self, nq_rows, device):
        return self.acnative.NNState(nq_rows, device, self.nn_cap or self.acnative.NN_CAP)

    def nn_scan(self, state, keys, keys_h16, queries, queries_h16, slack):
        self.acnative.nn_scan(state, keys, keys_h16, queries, queries_h16, slack)        # nothing is read back

    def nn_any_overflow(self, state):
        return self.acnative.nn_any_overflow(state)                                      # the search's one host read

    def nn_overflow_queries(self, state, queries):
        return self.acnative.OverflowQueries(state, queries)

    def nn_finalize(self, state, P):
        return self.acnative.nn_finalize(state, P)

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
    Pads to the largest block so one all_gather_into_tensor (one RCCL all-gather) moves everything.  With
    shard_bounds' layout (every non-empty shard but the last is full) the first sum(counts) rows of the
    receive buffer ARE the concatenation - no second copy."""
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
    total = sum(counts)
    full = 0
    while full < world and counts[full] == cap:
        full += 1
    if all(c == 0 for c in counts[full + 1:]):       # [cap]*full + [short] + [0]*rest: already contiguous
        return recv[:total]
    return torch.cat([recv[r * cap: r * cap + counts[r]] for r in range(world)], dim=0)


def select_bank(n_train, n_test, train_ratio, seed=0):
    """Indices of the normal-image bank.  The reference embeds the shuffled TRAIN loader and keeps the
    first int(train_ratio * N_test) images (examples/main.py:274-281: `len(Z)` is the TEST count, and the
    shuffle is unseeded); here the shuffle is a seeded permutation, so runs are reproducible."""
    m = min(n_train, int(train_ratio * n_test))
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(n_train, generator=g)[:m]


# ------------------------------------------------------------------ the ring of packed shards
class _Ring:
    """Shard exchange of the ring schedule: at step k every rank sends its own packed shard to rank - k and
    receives the shard of rank + k.  prefetch(k) posts the transfer, get(k) waits for it (the wait is a stream
    dependency under RCCL).  `depth` receive buffers: a small category posts EVERY step's transfer up front, so the
    shards travel over different xGMI links at the same time and under the local work of step 0; a large one
    (N = 10 000: 16 GB per shard) keeps two buffers and fetches step k + 2 when step k's buffer is free.
    Under a host backend (gloo: the CPU tests, and the tests that run several ranks on ONE GPU, which RCCL
    refuses) device tensors are staged through host memory."""

    def __init__(self, ops, local, sizes, rank, world, group, last_step, remote_bytes=REMOTE_BYTES):
        self.ops, self.local, self.sizes, self.rank, self.world, self.group = ops, local, sizes, rank, world, group
        self.cap = max(sizes) if sizes else 0
        self.last = last_step
        # bytes of one receive buffer from the SHAPE (hl32 row + its norm), not from the local tensors: a rank with an
        # empty shard must arrive at the same depth as the others
        per_buffer = max(1, (local.D * 4 + 4) * local.P * self.cap)
        self.depth = max(2, min(max(last_step, 1), remote_bytes // per_buffer))
        self.bufs = [None] * self.depth
        self.pending = {}
        self.stage = world > 1 and dist.get_backend(group) != "nccl" and local.tensors()[0].is_cuda
        self._host_local = None
        if self.depth >= last_step and last_step > 1:
            # every step's shard fits: ONE batch (one RCCL group) carries all of them, so the transfers to the different
            # peers run side by side, each over its own xGMI link.  Posted as one batch per step they would queue behind
            # one another on the communicator's stream: step k's shard would arrive after k transfer times, not one.
            self._post(list(range(1, last_step + 1)))
        else:
            for k in range(1, min(self.depth, last_step) + 1):
                self.prefetch(k)

    def _peer(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def prefetch(self, k):
        if k in self.pending or k <= 0 or k > self.last:
            return
        self._post([k])

    def _post(self, steps):
        """Post the exchanges of `steps` as ONE batch of point-to-point operations."""
        p2p, entries = [], []
        for k in steps:
            src, dst = (self.rank + k) % self.world, (self.rank - k) % self.world
            n_src = self.sizes[src]
            slot = k % self.depth
            if self.bufs[slot] is None and n_src:
                self.bufs[slot] = self.ops.empty_packed(self.cap, self.local)
            recv = self.bufs[slot].slice(0, n_src) if n_src else None
            send_t = self.local.tensors() if self.local.n else []          # an empty shard is neither sent nor received
            recv_t = recv.tensors() if n_src else []
            host_recv = None
            if self.stage:
                if self._host_local is None:
                    self._host_local = [t.cpu() for t in send_t]
                send_t = self._host_local
                host_recv = [torch.empty(t.shape, dtype=t.dtype) for t in recv_t]
            p2p += [dist.P2POp(dist.isend, t, self._peer(dst), self.group) for t in send_t]
            p2p += [dist.P2POp(dist.irecv, t, self._peer(src), self.group) for t in (host_recv if self.stage else recv_t)]
            entries.append((k, recv, host_recv))
        works = dist.batch_isend_irecv(p2p) if p2p else []
        batch = {"works": works, "waited": False}        # waited once, whichever step asks first (a gloo recv must not be waited twice)
        for k, recv, host_recv in entries:
            self.pending[k] = (batch, recv, host_recv)

    def get(self, k):
        """The shard of rank + k (None when that shard is empty).  Also posts the transfer that may now reuse the
        buffer of step k - 1 (whose kernels are already queued on the stream)."""
        self.prefetch(k)
        batch, recv, host_recv = self.pending.pop(k)
        if not batch["waited"]:
            for w in batch["works"]:
                w.wait()
            batch["waited"] = True
        if host_recv is not None:
            for d, h in zip(recv.tensors() if recv is not None else [], host_recv):
                d.copy_(h)
        self.prefetch(k - 1 + self.depth)
        return recv


def _slice_global(pk, base, rng):
    return pk.slice(rng[0] - base, rng[1] - base)


def alpha_weights_unsupervised(ops, pk_loc, bounds, rank, world, group, device, budget=MINBUF_BYTES, schedule="auto",
                               remote_bytes=REMOTE_BYTES):
    """w fp32 [n_loc, P] of the rank's images (utils.py:222-227 for every i of the shard)."""
    n = bounds[-1][1]
    lo, hi = bounds[rank]
    P = pk_loc.P
    counts = [b - a for a, b in bounds]
    if schedule == "auto":
        schedule = "ring"
    if schedule == "slices":
        rows = [c * P for c in counts]
        hl = all_gather_rows(pk_loc.hl, rows, group)
        sq = all_gather_rows(pk_loc.sq, rows, group)
        pk_all = type(pk_loc)(hl, sq, n, P, pk_loc.D)
        shift = ops.fixed_shift(ops.max_sq([pk_all], device), n)
        minbuf = ops.pair_block(pk_all, pk_all, part=rank, nparts=world)          # [N, N*P], +inf outside the slice
        if world > 1:
            mf = minbuf.view(torch.float32)
            dist.all_reduce(mf, op=dist.ReduceOp.MIN, group=group)
        acc = ops.new_acc((hi - lo) * P, MEAN, device)
        ops.pair_accumulate(minbuf[:, lo * P: hi * P], P, MEAN, lo, shift, acc)
        return ops.pair_finalize(acc, P, MEAN, n - 1, shift)

    m = ops.max_sq([pk_loc], device)
    if world > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    shift = ops.fixed_shift(m, n)
    acc = ops.new_acc(n * P, MEAN, device)                 # sums for EVERY image: remote queries' parts too
    c = block_images(P, budget)
    ring = _Ring(ops, pk_loc, counts, rank, world, group, last_step=world // 2, remote_bytes=remote_bytes)
    by_step = {}
    for item in unsup_schedule(bounds, rank, c):
        by_step.setdefault(item[1], []).append(item)
    for step in range(0, world // 2 + 1):
        remote, rbase = None, 0
        if step > 0:                                       # every rank takes part in every exchange, work or not
            remote = ring.get(step)
            rbase = bounds[(rank + step) % world][0]
        for item in by_step.get(step, []):
            if item[0] == "sym":
                a = item[2]
                A = _slice_global(pk_loc, lo, a)
                mb = ops.pair_block(A, A)
                ops.pair_accumulate(mb, P, MEAN, 0, shift, acc[a[0] * P: a[1] * P])
            else:
                a, b = item[2], item[3]
                A = _slice_global(pk_loc, lo, a)
                B = _slice_global(pk_loc, lo, b) if step == 0 else _slice_global(remote, rbase, b)
                if a[0] > b[0]:        # canonical orientation: the block with the LOWER image indices is the row
                    a, b, A, B = b, a, B, A   # operand, whichever rank computes it (a.b and b.a differ in the last bit)
                mb, mbt = ops.pair_block(A, B, two_way=True)
                ops.pair_accumulate(mb, P, MEAN, NO_SELF, shift, acc[b[0] * P: b[1] * P])     # keys A, queries B
                ops.pair_accumulate(mbt, P, MEAN, NO_SELF, shift, acc[a[0] * P: a[1] * P])    # keys B, queries A
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return ops.pair_finalize(acc[lo * P: hi * P], P, MEAN, n - 1, shift)


def alpha_weights_supervised(ops, pk_loc, pk_bank_loc, bank_bounds, rank, world, group, device, budget=MINBUF_BYTES,
                             remote_bytes=REMOTE_BYTES, prune=True):
    """w fp32 [n_loc, P]: min over every bank image (utils.py:230-237 for every i of the shard).  The bank
    shards go round the ring.  Default: the pruned EXACT nearest-neighbour search - per shard one MFMA launch on
    the fp16 high halves keeps, for every query patch, the bank patches within a proven error bound of the
    running minimum, and only those (about a dozen, mostly collected before the running minimum has settled) are
    measured exactly (fp64 on the split values).  A query patch with more near-ties than candidate slots is
    re-measured by the exhaustive kernel on its own.  prune=False (or an ops seam without the kernels, or D not a
    multiple of 64): exhaustive one-way split-fp16 blocks for everything, min into a uint32 accumulator."""
    P = pk_loc.P
    n_loc = pk_loc.n
    counts = [b - a for a, b in bank_bounds]
    m = sum(counts)
    if prune and getattr(ops, "nn_prune", False) and pk_loc.D % 64 == 0:
        mk, mq = ops.max_sq([pk_bank_loc], device), ops.max_sq([pk_loc], device)
        if world > 1:                               # the bound must hold for every (query, bank patch) pair of the run
            dist.all_reduce(mk, op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(mq, op=dist.ReduceOp.MAX, group=group)
        slack = ops.nn_slack(mk, mq)
        state = ops.nn_state(n_loc * P, device)
        q16 = ops.h16(pk_loc)
        ring = _Ring(ops, pk_bank_loc, counts, rank, world, group, last_step=world - 1, remote_bytes=remote_bytes)
        for k in range(world):                      # the whole ring is queued without a host read
            keys = pk_bank_loc if k == 0 else ring.get(k)
            if keys is None or keys.n == 0 or n_loc == 0:
                continue
            ops.nn_scan(state, keys, ops.h16(keys), pk_loc, q16, slack)
        # ONE host read per search: did any query patch, on any bank shard, have more near-ties than candidate slots
        # (e.g. a uniform background against a bank of the same background)?  On ordinary data: no, and this is
        # all.  Otherwise those queries - on every rank, hence the MAX all-reduce of the flag: the ring is collective
        # - meet the whole bank again through the exhaustive one-way split-fp16 blocks, on a compact copy, and take
        # that answer as it stands (bit-equal to prune=False).  A bank on which every query overflows therefore costs
        # one wasted first pass (a third of the exhaustive work) - the bound of the regression against prune=False.
        if world > 1:
            dist.all_reduce(state.overflow, op=dist.ReduceOp.MAX, group=group)
        if ops.nn_any_overflow(state):
            over = ops.nn_overflow_queries(state, pk_loc)
            ring = _Ring(ops, pk_bank_loc, counts, rank, world, group, last_step=world - 1, remote_bytes=remote_bytes)
            for k in range(world):
                over.scan(pk_bank_loc if k == 0 else ring.get(k), budget)
            over.commit(state)
        return ops.nn_finalize(state, P)
    acc = ops.new_acc(n_loc * P, MIN, device)
    ring = _Ring(ops, pk_bank_loc, counts, rank, world, group, last_step=world - 1, remote_bytes=remote_bytes)
    c = block_images(P, budget)
    for k in range(world):
        keys = pk_bank_loc if k == 0 else ring.get(k)
        if keys is None or keys.n == 0 or n_loc == 0:
            continue
        for q0, q1 in _subranges(0, n_loc, c):
            Q = pk_loc.slice(q0, q1)
            for k0, k1 in _subranges(0, keys.n, c):
                mb = ops.pair_block(keys.slice(k0, k1), Q)
                ops.pair_accumulate(mb, P, MIN, NO_SELF, None, acc[q0 * P: q1 * P])
    return ops.pair_finalize(acc, P, MIN, m, None)


def run_category(ops, images, supervised="unsupervised", tau=1.0, k=1, train_images=None, train_ratio=1.0,
                 group=None, bank_seed=0, want_pdist=True, want_host=True, timings=None, budget=MINBUF_BYTES,
                 schedule="auto", Z_loc=None, remote_bytes=REMOTE_BYTES, prune=True):
    """images: [N, 3, H, W] (every rank passes the same N; only the rank's shard is read).
    `tau` may be a list: the embeddings and the patch distances are computed once and every tau gets its
    own (alpha, X) in out["by_tau"][tau]; the top-level keys are those of the first tau.
    budget / schedule / remote_bytes: memory budget of one matrix of minima, the unsupervised schedule ("auto" =
    "ring", or "slices") and the memory the ring may spend on receive buffers - results do not depend on any.
    prune: supervised mode as a pruned exact nearest-neighbour search (default) or as exhaustive one-way blocks.
    Z_loc: the rank's embeddings from an earlier call on the same images (out["Z_loc"]): Z does not depend on
    the mode, so a run over several modes embeds the test set once (the reference re-embeds per mode and per tau,
    examples/main.py:346-366).
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
    if Z_loc is None:
        Z_loc = ops.embed(images[lo:hi])                               # [n_loc, P, D]
    elif Z_loc.shape[0] != hi - lo:
        raise ValueError("Z_loc holds %d images, this rank's shard has %d" % (Z_loc.shape[0], hi - lo))
    mark("embed", t0)
    device = Z_loc.device

    t0 = time.perf_counter()
    if supervised == "supervised":
        if train_images is None:
            raise ValueError("supervised alpha needs train_images")
        sel = select_bank(train_images.shape[0], n, train_ratio, bank_seed)
        bb = shard_bounds(len(sel), world)
        blo, bhi = bb[rank]
        bank_loc = ops.embed(train_images[sel[blo:bhi]])
        w_loc = alpha_weights_supervised(ops, ops.pack(Z_loc), ops.pack(bank_loc), bb, rank, world, group, device, budget,
                                         remote_bytes, prune)
        del bank_loc
    elif supervised == "unsupervised":
        w_loc = alpha_weights_unsupervised(ops, ops.pack(Z_loc), bounds, rank, world, group, device, budget, schedule,
                                           remote_bytes)
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
    out["w_loc"] = w_loc
    out["Z_loc"] = Z_loc
    return out
