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

What travels round the ring (round 4): with a backbone behind the seam, the tapped ViT feature maps of a shard
(`Taps`, 4.8 MB per image) instead of its packed embeddings (12.85 MB); the receiver re-aggregates and packs them -
the same kernel on the same bytes, so the sender's own packed rows bit for bit.

Several categories (`run_categories`, round 4; the reference loops over them one at a time, examples/main.py:346-366):
the rank's shards of ALL categories share full ViT batches, every ring is posted before the alpha work starts, the
steps that need no remote data run first for all categories, and the per-category collectives merge into one each;
or whole categories are dealt to ranks ("per_gpu").  Same numbers as the per-category loop, bit for bit.

`ops` is the compute seam: by default the HIP library (acnative + AnomalyClusteringCore); the
gloo tests (world 2-5) substitute a CPU stand-in to exercise the schedules and collectives.
"""
import math
import os
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
            # the pair {rank, src} meets from both sides at this step: the lower rank takes ALL of its images against
            # the first half of the HIGHER rank's, the higher rank its remaining images against all of the lower's.
            # (Round 6: the higher shard is the one that is split - rounds 2-5 split the lower - so that the lower
            # rank's block keeps its whole shard as one operand and joins the rectangle of its other steps:
            # merge_later_steps.)
            low, high = min(rank, src), max(rank, src)
            hlo, hhi = bounds[high]
            mid = hlo + (hhi - hlo + 1) // 2
            if rank == low:
                theirs = (hlo, mid)
            else:
                mine = (mid, hhi)
        for a in _subranges(mine[0], mine[1], c):
            for b in _subranges(theirs[0], theirs[1], c):
                yield ("two", k, a, b)


def merge_later_steps(items, c, P, budget=MINBUF_BYTES):
    """Fuse the two-way blocks of ring steps >= 1 (unsup_schedule's items, in step order) into as few launches as their
    shapes allow: a run of consecutive steps whose local block is the same, whose remote blocks follow one another in
    image order and lie on the same side of the local block (the orientation rule of _UnsupJob.run: the block with the
    lower image indices is the row operand) is ONE rectangle.  -> [("two", (k_first, k_last), a, b)].
    A rank of an 8-GPU run of an 83-image category then runs its steps 1-4 as one launch of 11 x 39 images instead of
    three of 11 x 11 and one of 6 x 11, each with its own ramp and tail; the ranks of the upper
    half, whose ring wraps round image 0, as two or three.  The fused block respects the launch limits (c images a
    side would have been split by unsup_schedule anyway; the matrices of minima stay inside the budget)."""
    out = []
    for it in items:
        _, k, a, b = it
        if out:
            _, (k1, k2), pa, pb = out[-1]
            fused = (pb[0], b[1])
            if (k == k2 + 1 and pa == a and pb[1] == b[0] and (pb[0] > a[0]) == (b[0] > a[0]) and fused[1] - fused[0] <= c
                    and (a[1] - a[0]) * (fused[1] - fused[0]) * P * 4 <= budget):
                out[-1] = ("two", (k1, k), a, fused)
                continue
        out.append(("two", (k, k), a, b))
    return out


# ------------------------------------------------------------------ ViT launch sizes by whole rounds of the chip
VIT_TILE_ROWS = 256            # row tile of the linear-layer kernels (csrc/gemm_hl5.h)
VIT_TAIL_WEIGHT = 0.18         # what an unfilled last round costs, as a share of a full one (measured: profiles/r04_vit_batch_sweep.txt)
VIT_LAUNCH_IMAGES = 1.5        # fixed cost of one ViT launch sequence, in images


def _vit_rounds(b, tokens, embed_dim, cus):
    """Tile rounds of the narrowest linear layer (proj / fc2: embed_dim columns) of a launch on b images; the wider layers
    (qkv: 3x, fc1: 4x the column tiles) fill whole rounds whenever this one does."""
    return -(-b * tokens // VIT_TILE_ROWS) * -(-embed_dim // 256) / float(cus)


def vit_batch_cost(b, tokens, embed_dim, cus=256):
    """Relative cost (in images) of one ViT launch on b images: b, plus the idle share of the last tile round.  A launch on
    84 images needs 9.07 rounds of 256 tiles for qkv (3.02 for proj): the chip waits for the 0.07 - measured +5.3 % per
    image against 83 images (8.96 rounds); 111 / 138 / 166 / 194 / 222 / 250 images sit just below whole rounds, too."""
    if b <= 0:
        return 0.0
    r = _vit_rounds(b, tokens, embed_dim, cus)
    idle = math.ceil(r - 1e-9) - r
    return b * (1.0 + VIT_TAIL_WEIGHT * max(idle, 0.0) / r) + VIT_LAUNCH_IMAGES


def vit_batch_plan(total, limit, tokens, embed_dim, cus=256):
    """Sizes of the ViT launches over `total` images, none above `limit`: launches that fill whole tile rounds of the chip
    (the largest such size <= limit, and smaller ones where they save a ragged remainder), the one remainder last.  An
    image's embedding does not depend on its batch (tests: bitwise), so this only moves time: 1725 images in launches of
    128 cost 445 us per image, in launches of 250 / 222 / ... 431."""
    total, limit = int(total), max(1, int(limit))
    if total <= limit:
        return [total] if total else []
    good = []
    k = 1
    while True:
        b = (cus * k // -(-embed_dim // 256)) * VIT_TILE_ROWS // tokens     # largest b with rounds(b) <= k
        if b > limit:
            break
        if b >= 1 and (not good or b != good[-1]):
            good.append(b)
        k += 1
    if not good:
        good = [limit]
    cost = {g: vit_batch_cost(g, tokens, embed_dim, cus) for g in good}
    INF = float("inf")
    best, take = [0.0] + [INF] * total, [0] * (total + 1)       # launches of good sizes only (unbounded knapsack)
    for t in range(1, total + 1):
        for g in good:
            if g <= t and best[t - g] + cost[g] < best[t]:
                best[t], take[t] = best[t - g] + cost[g], g
    pick, rest = None, 0
    for r in range(0, limit + 1):                               # ... plus ONE launch of any size
        if r <= total and best[total - r] < INF:
            c = best[total - r] + vit_batch_cost(r, tokens, embed_dim, cus)
            if pick is None or c < pick:
                pick, rest = c, r
    plan, t = [], total - rest
    while t > 0:
        plan.append(take[t])
        t -= take[t]
    plan.sort(reverse=True)
    plan += [rest] if rest else []
    naive = [limit] * (total // limit) + ([total % limit] if total % limit else [])
    if sum(vit_batch_cost(b, tokens, embed_dim, cus) for b in naive) < pick:      # never worse than plain batches of `limit`
        return naive
    return plan


class Taps:
    """The tapped feature maps of a block of images ([n, 785, 768] x 2 for ViT-B/8: 4.8 MB per image) - what goes round
    the ring instead of the packed embeddings (12.85 MB per image): the receiver runs the aggregation kernel on them
    (17.7 MB of HBM traffic per image, microseconds) and packs the result, bit-identical to what the sender holds.
    Same surface as a packed block as far as the ring is concerned: n, slice(), tensors().
    sq (round 6): the squared norms [n, P] of the images' packed rows, when the sender has them (with_norms: 3 KB per
    image next to 4.8 MB) - the receiver then writes the hl32 rows straight from the aggregation kernel and never forms
    the fp32 Z of a block it only multiplies (HipOps.taps_to_packed)."""

    def __init__(self, feats, n, sq=None):
        self.feats, self.n, self.sq = list(feats), int(n), sq

    def with_norms(self, sq_rows, P):
        """These taps + the squared norms of their packed rows (fp32 [n * P], Packed.sq of the same images).  An empty
        shard gets an empty [0, P]: every rank's taps must have the same tensors (the ring's receive buffers mirror them)."""
        return Taps(self.feats, self.n, sq_rows.reshape(self.n, P))

    def slice(self, lo, hi):
        return Taps([f[lo:hi] for f in self.feats], hi - lo, None if self.sq is None else self.sq[lo:hi])

    def tensors(self):
        return self.feats if self.sq is None else self.feats + [self.sq]

    def bytes_per_image(self):
        return sum(int(np.prod(f.shape[1:])) * f.element_size() for f in self.tensors())

    def empty_like(self, n):
        return Taps([torch.empty((n,) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device) for f in self.feats], n,
                    None if self.sq is None else torch.empty((n,) + tuple(self.sq.shape[1:]), dtype=self.sq.dtype, device=self.sq.device))

    @property
    def device(self):
        return self.feats[0].device


class HipOps:
    """Compute seam backed by libac_hip.so.  Every method raises if the library or the GPU is missing."""

    def __init__(self, core, batch_size=32):
        import acnative
        self.acnative = acnative
        self.core = core
        self.batch_size = batch_size          # the LARGEST ViT launch; the sizes actually used: batches()
        self._zshape = None
        self._tapshapes = None
        self._plans = {}
        # what a multi-rank run sends round the ring: the taps (2.7x fewer bytes) when there is a backbone to tap
        # (AC_RING_PAYLOAD=packed keeps the packed shards selectable: A/B on the first multi-GPU node)
        self.ring_payload = "taps" if (core is not None and os.environ.get("AC_RING_PAYLOAD", "taps") == "taps") else None

    def batches(self, total):
        """The launch sizes over `total` images: for a token backbone (ViT) those that fill whole tile rounds of the chip
        (vit_batch_plan); otherwise plain batches of batch_size."""
        bb = getattr(self.core, "backbone", None)
        tokens = getattr(getattr(bb, "patch_embed", None), "num_patches", None)
        dim = getattr(bb, "embed_dim", None)
        if not tokens or not dim or total <= self.batch_size or os.environ.get("AC_VIT_BATCH_PLAN") == "naive":
            return self.batch_size                      # (AC_VIT_BATCH_PLAN=naive: plain batches, for a same-box A/B)
        cus = torch.cuda.get_device_properties(self.core.device).multi_processor_count
        key = (int(total), self.batch_size, tokens, dim, cus)
        if key not in self._plans:
            self._plans[key] = vit_batch_plan(total, self.batch_size, tokens + 1, dim, cus)
        return self._plans[key]

    def _probe(self):
        probe = torch.ones((1,) + tuple(self.core.input_shape), device=self.core.device)
        Zs, taps = self.core.embed_device_many([probe], batch_size=1, want_taps=True)
        self._zshape = tuple(Zs[0].shape[1:])
        self._tapshapes = [(tuple(t.shape[1:]), t.dtype) for t in taps[0]]

    def _empty(self, want_taps):
        if self._zshape is None or (want_taps and self._tapshapes is None):
            self._probe()
        Z = torch.empty((0,) + self._zshape, dtype=torch.float32, device=self.core.device)
        if not want_taps:
            return Z
        return Z, Taps([torch.empty((0,) + sh, dtype=dt, device=self.core.device) for sh, dt in self._tapshapes], 0)

    def embed(self, images, want_taps=False):
        """Z fp32 [n, P, D] of the images; want_taps: (Z, Taps of the same images) for the ring."""
        if images.shape[0] == 0:                     # empty shard (fewer images than ranks)
            return self._empty(want_taps)
        if want_taps:
            Zs, taps = self.embed_many([images], want_taps=True)
            return Zs[0], taps[0]
        Z = self.core.embed_device(images, batch_size=self.batches(images.shape[0]))
        self._zshape = tuple(Z.shape[1:])
        return Z

    def embed_many(self, image_sets, want_taps=False):
        """Several image sets (the rank's shards of several categories) through the ViT as ONE stream of full
        batches: [Z of set 0, Z of set 1, ...], views of one buffer.  A rank of an 8-GPU run holds 5-21 images of
        an MVTec category - a fifth of a ViT batch; the sets of all 15 categories together fill whole batches."""
        if sum(int(s.shape[0]) for s in image_sets) == 0:
            res = [self._empty(want_taps) for _ in image_sets]
            return ([r[0] for r in res], [r[1] for r in res]) if want_taps else res
        out = self.core.embed_device_many(image_sets, batch_size=self.batches(sum(int(x.shape[0]) for x in image_sets)),
                                          want_taps=want_taps)
        Zs = out[0] if want_taps else out
        self._zshape = tuple(Zs[0].shape[1:])
        if want_taps:
            self._tapshapes = [(tuple(t.shape[1:]), t.dtype) for t in out[1][0]]
            return Zs, [Taps(feats, int(z.shape[0])) for feats, z in zip(out[1], Zs)]
        return Zs

    def taps_to_packed(self, taps):
        """A received block of taps -> its packed embeddings (aggregate + pack: the sender's own Z, bit for bit).  When the
        sender's norms came along (Taps.sq) and the shape is the fused kernel's, the aggregation writes the hl32 rows itself
        and the norms are the sender's: the same bytes without the fp32 Z of the block (25.7 MB of HBM traffic per image)."""
        if taps.n == 0:
            return self.pack(self._empty(False))
        if taps.sq is not None and os.environ.get("AC_RING_FUSED_DECODE", "1") != "0":
            P, D = self.zshape()
            hl = torch.empty((taps.n * P, D * 4), dtype=torch.uint8, device=taps.device)
            if self.core.aggregate_taps_hl32(taps.feats, hl):
                return self.acnative.Packed(hl, taps.sq.reshape(-1), taps.n, P, D)
        return self.pack(self.core.aggregate_taps(taps.feats))

    def taps_with_norms(self, taps, packed):
        """What a ring of taps sends: the taps + the norms of the same images' packed rows (see Taps.sq)."""
        if taps.sq is not None:
            return taps
        if taps.n != packed.n:
            raise ValueError("taps of %d images, packed rows of %d" % (taps.n, packed.n))
        return taps.with_norms(packed.sq, packed.P)

    # ---- a category too large for Z next to its packed copy keeps the taps instead of Z (run_category's `lean`)
    LEAN_CHUNK = 128          # images aggregated at a time when Z is rebuilt from taps (1.6 GB of Z)
    LEAN_FRACTION = 0.5       # lean when Z + packed of the rank's images would take more than this share of the free HBM

    def zshape(self):
        """(P, D) of one image's Z."""
        if self._zshape is None:
            probe = torch.ones((1,) + tuple(self.core.input_shape), device=self.core.device)
            self._zshape = tuple(self.core.embed_device(probe, batch_size=1).shape[1:])
        return self._zshape

    def never_lean(self, n_images):
        """True when Z + packed of n_images is so small (an eighth of the device's TOTAL memory: 36 GB on an MI355X, 1 400
        images) that no rank of the job can want the lean mode whatever its free HBM is.  A function of the shapes and of
        the device model only - every rank of a run on one kind of GPU gets the same answer without asking the others
        (run_categories: no all-reduce and no host read in a steady-state step of configs 3 / 4 on several ranks)."""
        if self.core is None:
            return True
        P, D = self.zshape()
        return n_images * 2 * P * D * 4 <= torch.cuda.get_device_properties(self.core.device).total_memory // 8

    def wants_lean(self, n_images):
        """Lean when Z + packed of n_images would not fit comfortably - and the backbone's taps can be kept at all (token
        maps; the channels-last views of the CNN branch cannot, and its categories are small)."""
        if self.core is None:                 # the engine alone (no backbone behind the seam): there are no taps to keep
            return False
        P, D = self.zshape()
        free, _ = torch.cuda.mem_get_info(self.core.device)
        if n_images * 2 * P * D * 4 <= self.LEAN_FRACTION * free:
            return False
        if self._tapshapes is None:
            try:
                self._probe()
            except RuntimeError:
                return False
        return True

    def embed_taps(self, images):
        """Taps of the images, Z never materialised as a whole (25.7 -> 17.7 MB of HBM per image with the packed copy)."""
        if images.shape[0] == 0:
            return self._empty(True)[1]
        feats = self.core.embed_taps_device(images, batch_size=self.batches(images.shape[0]))
        self._tapshapes = [(tuple(t.shape[1:]), t.dtype) for t in feats]
        return Taps(feats, int(images.shape[0]))

    def pack_taps(self, taps):
        """Taps -> Packed, LEAN_CHUNK images at a time: aggregate into a scratch Z, pack into place.  Bit-identical to
        pack(embed(images)): the same two kernels on the same bytes."""
        P, D = self.zshape()
        pk = self.acnative.Packed.empty(taps.n, P, D, taps.device)
        for s in range(0, taps.n, self.LEAN_CHUNK):
            e = min(taps.n, s + self.LEAN_CHUNK)
            self.acnative.pack_images(self.core.aggregate_taps([f[s:e] for f in taps.feats]), out=pk.slice(s, e))
        return pk

    def weighted_avg_taps(self, alpha_f32, taps):
        """X = alpha . Z with Z rebuilt from the taps chunk by chunk (main.py:292-296 on a category whose Z was dropped)."""
        P, D = self.zshape()
        X = torch.empty((taps.n, D), dtype=torch.float32, device=taps.device)
        if taps.n == 0:                                        # a rank with an empty image shard
            return X
        alpha_f32 = alpha_f32.reshape(taps.n, -1)
        for s in range(0, taps.n, self.LEAN_CHUNK):
            e = min(taps.n, s + self.LEAN_CHUNK)
            self.acnative.weighted_avg(alpha_f32[s:e], self.core.aggregate_taps([f[s:e] for f in taps.feats]), out=X[s:e])
        return X

    # ---- packed blocks and the pair kernels (see acnative for the semantics)
    def pack(self, Z):
        return self.acnative.pack_images(Z)

    def empty_packed(self, n, like):
        return self.acnative.Packed.empty(n, like.P, like.D, like.hl.device)

    def max_sq(self, packed_list, device, out=None):
        return self.acnative.max_sq(packed_list, device, out)

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

    def nn_state(self, nq_rows, device, overflow=None):
        return self.acnative.NNState(nq_rows, device, self.nn_cap or self.acnative.NN_CAP, overflow)

    def nn_scan(self, state, keys, keys_h16, queries, queries_h16, slack):
        self.acnative.nn_scan(state, keys, keys_h16, queries, queries_h16, slack)        # nothing is read back

    def nn_any_overflow(self, state):
        return self.acnative.nn_any_overflow(state)                                      # the search's one host read

    def nn_overflow_queries(self, state, queries):
        return self.acnative.OverflowQueries(state, queries)

    def nn_finalize(self, state, P):
        return self.acnative.nn_finalize(state, P)

    def alpha_softmax(self, w, tau, k):
        """-> (alpha fp64 [n, P], its fp32 copy: main.py:294's `.float()`), both written by the one kernel."""
        return self.acnative.alpha_softmax(w, tau, k, want_f32=True)

    def alpha_average(self, n, P, device):
        """main.py:290-291: ones / P, as fp32 [n, P] (the bit pattern of float32(1/P), filled by the library)."""
        return self.acnative.alpha_average(n, P, device)

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


def all_gather_rows(local, counts, group=None, rot=0):
    """Concatenate per-rank row blocks of unequal length (counts[v] rows on the rank whose VIRTUAL rank is v = (rank +
    rot) % world: see run_category's `rot`) on every rank, in virtual-rank (= image) order.
    Pads to the largest block so one all_gather_into_tensor (one RCCL all-gather) moves everything.  With
    shard_bounds' layout (every non-empty shard but the last is full) the first sum(counts) rows of the
    receive buffer ARE the concatenation - no second copy.  len(counts) == 1: a run on one rank (or a rank
    working a whole category on its own inside a larger job) - nothing to gather."""
    if len(counts) == 1:
        return local
    rank, world = _world(group)
    if world != len(counts):
        raise ValueError("all_gather_rows: %d counts for a group of %d ranks" % (len(counts), world))
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
    if rot % world == 0 and all(c == 0 for c in counts[full + 1:]):   # [cap]*full + [short] + [0]*rest: already contiguous
        return recv[:total]
    # block v of the result sits in the receive slot of the REAL rank (v - rot) % world
    return torch.cat([recv[((v - rot) % world) * cap: ((v - rot) % world) * cap + counts[v]] for v in range(world)], dim=0)


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

    def __init__(self, ops, local, sizes, rank, world, group, last_step, remote_bytes=REMOTE_BYTES, post=True, decode=None,
                 rot=0):
        """local: the rank's own block as it travels - a packed block, or its Taps with decode = ops.taps_to_packed
        (get() then hands out the packed block the consumer wants).  rank / sizes are in VIRTUAL rank space (run_category's
        `rot`): the process behind virtual rank v is group rank (v - rot) % world."""
        self.ops, self.local, self.sizes, self.rank, self.world, self.group = ops, local, sizes, rank, world, group
        self.rot = rot
        self.decode = decode
        self.cap = max(sizes) if sizes else 0
        self.last = last_step
        # bytes of one receive buffer from the SHAPE (hl32 row + its norm; or the tap maps), not from the local tensors:
        # a rank with an empty shard must arrive at the same depth as the others
        per_image = local.bytes_per_image() if hasattr(local, "bytes_per_image") else (local.D * 4 + 4) * local.P
        per_buffer = max(1, per_image * self.cap)
        self.depth = max(2, min(max(last_step, 1), remote_bytes // per_buffer))
        self.bufs = [None] * self.depth
        self.pending = {}
        # every step posted up front: ONE receive store, the shards side by side in STEP order, so that a run of steps is
        # one block (get_run)
        self.contig = False
        self.store, self.off = None, {}
        self.stage = world > 1 and dist.get_backend(group) != "nccl" and local.tensors()[0].is_cuda
        self._host_local = None
        # depth is a function of the category's SHAPE and the caller's budget only, so every rank arrives at the same
        # value and takes the same branch below; AC_RING_CHECK=1 proves it with an all-gather (diagnostic)
        if world > 1 and os.environ.get("AC_RING_CHECK") == "1":
            seen = [None] * world
            dist.all_gather_object(seen, (self.depth, last_step), group=group)
            if len(set(seen)) != 1:
                raise RuntimeError("ring depth / last step differ across ranks: %r" % (seen,))
        if post and last_step >= 1:
            self.post()

    def post(self):
        """Post the first transfers: all of them as ONE batch when every step's shard has a buffer, else the first
        `depth` steps.  AC_RING_POST=per_step keeps the older one-batch-per-step posting selectable (A/B on the first
        multi-GPU node: transfers to different peers then queue behind one another on the communicator's stream)."""
        last_step = self.last
        if self.depth >= last_step and last_step > 1 and os.environ.get("AC_RING_POST", "batch") != "per_step":
            # every step's shard fits: ONE batch (one RCCL group) carries all of them, so the transfers to the different
            # peers run side by side, each over its own xGMI link.  Posted as one batch per step they would queue behind
            # one another on the communicator's stream: step k's shard would arrive after k transfer times, not one.
            steps = list(range(1, last_step + 1))
            slots = [k % self.depth for k in steps]
            if len(set(slots)) != len(slots):
                raise RuntimeError("ring: batched steps %r share receive buffers %r" % (steps, slots))
            self._lay_out(steps)
            self._post(steps)
        else:
            for k in range(1, min(self.depth, last_step) + 1):
                self.prefetch(k)

    def _lay_out(self, steps):
        """One receive store for all of `steps`: the shard of step k at image offset off[k], no gaps."""
        total = 0
        for k in steps:
            self.off[k] = total
            total += self.sizes[(self.rank + k) % self.world]
        self.contig = True
        if total:
            self.store = (self.local.empty_like(total) if hasattr(self.local, "empty_like")
                          else self.ops.empty_packed(total, self.local))

    def _peer(self, v):
        r = (v - self.rot) % self.world
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
            if self.contig:
                recv = self.store.slice(self.off[k], self.off[k] + n_src) if n_src else None
            else:
                if self.bufs[slot] is None and n_src:
                    self.bufs[slot] = (self.local.empty_like(self.cap) if hasattr(self.local, "empty_like")
                                       else self.ops.empty_packed(self.cap, self.local))
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
        if recv is not None and self.decode is not None:
            recv = self.decode(recv)          # new tensors: the receive buffer itself is free again once this is queued
        self.prefetch(k - 1 + self.depth)
        return recv

    def get_run(self, k1, k2):
        """The shards of steps k1 .. k2 as ONE block, in step order (None when all of them are empty); decoded by one
        call.  Only when every step was posted up front into the one store (self.contig)."""
        if k1 == k2:
            return self.get(k1)
        if not self.contig:
            raise RuntimeError("ring: get_run needs every step posted into one store")
        n = 0
        for k in range(k1, k2 + 1):
            batch, recv, host_recv = self.pending.pop(k)
            if not batch["waited"]:
                for w in batch["works"]:
                    w.wait()
                batch["waited"] = True
            if host_recv is not None:
                for d, h in zip(recv.tensors() if recv is not None else [], host_recv):
                    d.copy_(h)
            n += recv.n if recv is not None else 0
        if n == 0:
            return None
        blk = self.store.slice(self.off[k1], self.off[k1] + n)
        return self.decode(blk) if self.decode is not None else blk


def _slice_global(pk, base, rng):
    return pk.slice(rng[0] - base, rng[1] - base)


class _UnsupJob:
    """The unsupervised weights of ONE category on one rank (utils.py:222-227 for every image of the shard), cut into
    the phases that run_category walks in order and run_categories interleaves across categories:

        __init__   squared-norm maximum of the local shard   -> self.m (fp32 [1]; MAX all-reduce it across ranks)
        prepare    fixed-point scale from m, int64 sums      -> self.acc ([N*P]; SUM all-reduce it after the last step)
        post       the ring: every step's shard transfer posted (one batch)
        run(step)  the block launches of one ring step; step 0 needs no remote data
        finish     w fp32 [n_loc, P]

    `m` and `acc` may be views of tensors the caller owns (run_categories keeps ONE vector of maxima and ONE
    accumulator for all categories, so that each costs one collective instead of one per category)."""

    def __init__(self, ops, pk_loc, bounds, rank, world, group, device, budget=MINBUF_BYTES, remote_bytes=REMOTE_BYTES,
                 m_out=None, acc=None, send=None, rot=0):
        """send: the Taps of the local shard - what travels instead of pk_loc when the seam can re-aggregate them.
        rank is the VIRTUAL rank (run_category's `rot`)."""
        self.ops, self.pk, self.bounds, self.rank, self.world, self.group, self.device = ops, pk_loc, bounds, rank, world, group, device
        self.send, self.rot = send, rot
        self.n = bounds[-1][1]
        self.lo, self.hi = bounds[rank]
        self.P = pk_loc.P
        self.counts = [b - a for a, b in bounds]
        self.budget, self.remote_bytes = budget, remote_bytes
        self.m = ops.max_sq([pk_loc], device) if m_out is None else ops.max_sq([pk_loc], device, out=m_out)
        self.acc = acc
        self.ring = None
        self.last_step = world // 2

    def prepare(self):
        ops, P = self.ops, self.P
        self.shift = ops.fixed_shift(self.m, self.n)
        if self.acc is None:
            self.acc = ops.new_acc(self.n * P, MEAN, self.device)          # sums for EVERY image: remote queries' parts too
        self.by_step = {}
        for item in unsup_schedule(self.bounds, self.rank, block_images(P, self.budget)):
            self.by_step.setdefault(item[1], []).append(item)

    def post(self):
        send = self.send
        if send is not None and hasattr(self.ops, "taps_with_norms"):
            send = self.ops.taps_with_norms(send, self.pk)        # the packed rows' norms travel with the taps
        self.ring = _Ring(self.ops, send if send is not None else self.pk, self.counts, self.rank, self.world,
                          self.group, last_step=self.last_step, remote_bytes=self.remote_bytes,
                          decode=self.ops.taps_to_packed if send is not None else None, rot=self.rot)

    def run(self, step):
        ops, P, acc, shift, lo = self.ops, self.P, self.acc, self.shift, self.lo
        remote, rbase = None, 0
        if step > 0:                                       # every rank takes part in every exchange, work or not
            remote = self.ring.get(step)
            rbase = self.bounds[(self.rank + step) % self.world][0]
        for item in self.by_step.get(step, []):
            if item[0] == "sym":
                a = item[2]
                A = _slice_global(self.pk, lo, a)
                mb = ops.pair_block(A, A)
                ops.pair_accumulate(mb, P, MEAN, 0, shift, acc[a[0] * P: a[1] * P])
            else:
                a, b = item[2], item[3]
                A = _slice_global(self.pk, lo, a)
                B = _slice_global(self.pk, lo, b) if step == 0 else _slice_global(remote, rbase, b)
                if a[0] > b[0]:        # canonical orientation: the block with the LOWER image indices is the row
                    a, b, A, B = b, a, B, A   # operand, whichever rank computes it (a.b and b.a differ in the last bit)
                mb, mbt = ops.pair_block(A, B, two_way=True)
                ops.pair_accumulate(mb, P, MEAN, NO_SELF, shift, acc[b[0] * P: b[1] * P])     # keys A, queries B
                ops.pair_accumulate(mbt, P, MEAN, NO_SELF, shift, acc[a[0] * P: a[1] * P])    # keys B, queries A

    def run_later(self):
        """Every step >= 1.  With all of the ring's shards in one store (a small category: everything was posted up front)
        the steps' blocks run as the few rectangles merge_later_steps makes of them; otherwise step by step.
        AC_RING_MERGE=0 keeps the step-by-step order selectable (A/B on the first multi-GPU node)."""
        last = self.last_step
        if last < 1:
            return
        if not (getattr(self.ring, "contig", False) and last > 1 and os.environ.get("AC_RING_MERGE", "1") != "0"):
            for step in range(1, last + 1):
                self.run(step)
            return
        items = [it for step in range(1, last + 1) for it in self.by_step.get(step, [])]
        covered = set()
        for _, (k1, k2), a, b in merge_later_steps(items, block_images(self.P, self.budget), self.P, self.budget):
            if (k1, k2) in covered:          # (several sub-blocks of one step: the shard is fetched once)
                remote = self._last_remote
            else:
                remote = self._last_remote = self.ring.get_run(k1, k2)
                covered.add((k1, k2))
            self._two_way(a, b, remote, self.bounds[(self.rank + k1) % self.world][0])
        self._last_remote = None
        for k in range(1, last + 1):          # every rank takes part in every exchange, work or not
            if k in self.ring.pending:
                self.ring.get(k)

    def _two_way(self, a, b, remote, rbase):
        ops, P, acc, shift, lo = self.ops, self.P, self.acc, self.shift, self.lo
        A = _slice_global(self.pk, lo, a)
        B = _slice_global(remote, rbase, b)
        if a[0] > b[0]:                       # canonical orientation (see run)
            a, b, A, B = b, a, B, A
        mb, mbt = ops.pair_block(A, B, two_way=True)
        ops.pair_accumulate(mb, P, MEAN, NO_SELF, shift, acc[b[0] * P: b[1] * P])
        ops.pair_accumulate(mbt, P, MEAN, NO_SELF, shift, acc[a[0] * P: a[1] * P])

    def finish(self):
        P = self.P
        self.ring = None
        return self.ops.pair_finalize(self.acc[self.lo * P: self.hi * P], P, MEAN, self.n - 1, self.shift)


def alpha_weights_unsupervised(ops, pk_loc, bounds, rank, world, group, device, budget=MINBUF_BYTES, schedule="auto",
                               remote_bytes=REMOTE_BYTES, send=None, rot=0):
    """w fp32 [n_loc, P] of the rank's images (utils.py:222-227 for every i of the shard)."""
    n = bounds[-1][1]
    lo, hi = bounds[rank]
    P = pk_loc.P
    counts = [b - a for a, b in bounds]
    if schedule == "auto":
        schedule = "ring"
    if schedule == "slices":
        rows = [c * P for c in counts]
        hl = all_gather_rows(pk_loc.hl, rows, group, rot)
        sq = all_gather_rows(pk_loc.sq, rows, group, rot)
        pk_all = type(pk_loc)(hl, sq, n, P, pk_loc.D)
        shift = ops.fixed_shift(ops.max_sq([pk_all], device), n)
        minbuf = ops.pair_block(pk_all, pk_all, part=rank, nparts=world)          # [N, N*P], +inf outside the slice
        if world > 1:
            mf = minbuf.view(torch.float32)
            dist.all_reduce(mf, op=dist.ReduceOp.MIN, group=group)
        acc = ops.new_acc((hi - lo) * P, MEAN, device)
        ops.pair_accumulate(minbuf[:, lo * P: hi * P], P, MEAN, lo, shift, acc)
        return ops.pair_finalize(acc, P, MEAN, n - 1, shift)

    job = _UnsupJob(ops, pk_loc, bounds, rank, world, group, device, budget, remote_bytes, send=send, rot=rot)
    if world > 1:
        dist.all_reduce(job.m, op=dist.ReduceOp.MAX, group=group)
    job.prepare()
    job.post()
    job.run(0)
    job.run_later()
    if world > 1:
        dist.all_reduce(job.acc, op=dist.ReduceOp.SUM, group=group)
    return job.finish()


class _SupJob:
    """The supervised weights of ONE category on one rank as a pruned EXACT nearest-neighbour search, in phases (see
    _UnsupJob): __init__ (maxima mk, mq: MAX all-reduce both) -> prepare -> post (the bank shards' ring) -> run(k) for
    k = 0 .. world-1 (k = 0: the rank's own bank shard) -> the overflow flag `state.overflow` (MAX all-reduce, ONE
    host read) -> fix() when it is set (collective: every rank or none) -> finish."""

    def __init__(self, ops, pk_loc, pk_bank_loc, bank_bounds, rank, world, group, device, budget=MINBUF_BYTES,
                 remote_bytes=REMOTE_BYTES, m_out=None, overflow=None, send=None, rot=0):
        """send: the Taps of the local BANK shard (travels instead of pk_bank_loc, see _UnsupJob); rank: VIRTUAL rank."""
        self.ops, self.pk, self.bank, self.rank, self.world, self.group, self.device = ops, pk_loc, pk_bank_loc, rank, world, group, device
        self.send, self.rot = send, rot
        self.P, self.n_loc = pk_loc.P, pk_loc.n
        self.counts = [b - a for a, b in bank_bounds]
        self.budget, self.remote_bytes = budget, remote_bytes
        if m_out is None:
            self.mk, self.mq = ops.max_sq([pk_bank_loc], device), ops.max_sq([pk_loc], device)
        else:                                       # views of the caller's vector: one MAX all-reduce for all categories
            self.mk, self.mq = ops.max_sq([pk_bank_loc], device, out=m_out[0:1]), ops.max_sq([pk_loc], device, out=m_out[1:2])
        self._overflow = overflow
        self.last_step = world - 1
        self.ring = None

    def prepare(self):
        ops = self.ops
        self.slack = ops.nn_slack(self.mk, self.mq)          # the bound must hold for every (query, bank patch) pair of the run
        self.state = (ops.nn_state(self.n_loc * self.P, self.device) if self._overflow is None
                      else ops.nn_state(self.n_loc * self.P, self.device, overflow=self._overflow))
        self.q16 = ops.h16(self.pk)

    def _new_ring(self):
        send = self.send
        if send is not None and hasattr(self.ops, "taps_with_norms"):
            send = self.ops.taps_with_norms(send, self.bank)       # the packed rows' norms travel with the taps
        return _Ring(self.ops, send if send is not None else self.bank, self.counts, self.rank, self.world,
                     self.group, last_step=self.last_step, remote_bytes=self.remote_bytes,
                     decode=self.ops.taps_to_packed if send is not None else None, rot=self.rot)

    def post(self):
        self.ring = self._new_ring()

    def run(self, k):
        keys = self.bank if k == 0 else self.ring.get(k)
        if keys is None or keys.n == 0 or self.n_loc == 0:
            return
        self.ops.nn_scan(self.state, keys, self.ops.h16(keys), self.pk, self.q16, self.slack)

    def run_later(self):
        """Every step >= 1: the bank shards of the other ranks.  With all of them in one store (every step posted up
        front) they are ONE block of keys and one scan - the search is exact and takes the min over all bank patches,
        so neither the order of the shards nor their grouping changes a bit of the result.  AC_RING_MERGE=0: shard by
        shard."""
        last = self.last_step
        if last < 1:
            return
        if getattr(self.ring, "contig", False) and last > 1 and os.environ.get("AC_RING_MERGE", "1") != "0":
            keys = self.ring.get_run(1, last)
            if keys is not None and keys.n and self.n_loc:
                self.ops.nn_scan(self.state, keys, self.ops.h16(keys), self.pk, self.q16, self.slack)
            return
        for k in range(1, last + 1):
            self.run(k)

    def fix(self):
        """Some query patch, on some rank, had more near-ties than candidate slots (e.g. a uniform background against a
        bank of the same background): the marked queries of EVERY rank meet the whole bank again through the exhaustive
        one-way split-fp16 blocks, on a compact copy, and take that answer as it stands (bit-equal to prune=False).  A
        rank without marked queries (or without images) still carries the shards round the ring."""
        over = self.ops.nn_overflow_queries(self.state, self.pk)
        ring = self._new_ring()
        for k in range(self.world):
            over.scan(self.bank if k == 0 else ring.get(k), self.budget)
        over.commit(self.state)

    def finish(self):
        self.ring = None
        return self.ops.nn_finalize(self.state, self.P)


def _can_prune(ops, pk_loc, prune):
    return bool(prune and getattr(ops, "nn_prune", False) and pk_loc.D % 64 == 0)


def alpha_weights_supervised(ops, pk_loc, pk_bank_loc, bank_bounds, rank, world, group, device, budget=MINBUF_BYTES,
                             remote_bytes=REMOTE_BYTES, prune=True, send=None, rot=0):
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
    if _can_prune(ops, pk_loc, prune):
        job = _SupJob(ops, pk_loc, pk_bank_loc, bank_bounds, rank, world, group, device, budget, remote_bytes, send=send, rot=rot)
        if world > 1:
            dist.all_reduce(job.mk, op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(job.mq, op=dist.ReduceOp.MAX, group=group)
        job.prepare()
        job.post()
        job.run(0)                                  # the whole ring is queued without a host read
        job.run_later()
        # ONE host read per search: did any query patch, on any bank shard, have more near-ties than candidate slots?
        # On ordinary data: no, and this is all.  Otherwise those queries - on every rank, hence the MAX all-reduce of
        # the flag: the ring is collective - take the exhaustive path (fix).  A bank on which every query overflows
        # therefore costs one wasted first pass (a third of the exhaustive work) - the bound of the regression against
        # prune=False.
        if world > 1:
            dist.all_reduce(job.state.overflow, op=dist.ReduceOp.MAX, group=group)
        if ops.nn_any_overflow(job.state):
            job.fix()
        return job.finish()
    acc = ops.new_acc(n_loc * P, MIN, device)
    ring = _Ring(ops, send if send is not None else pk_bank_loc, counts, rank, world, group, last_step=world - 1,
                 remote_bytes=remote_bytes, decode=ops.taps_to_packed if send is not None else None, rot=rot)
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


def _flag_device(ops, images):
    """Where a one-word flag of a collective lives: the seam's device (gloo stand-ins without a core: the images')."""
    core = getattr(ops, "core", None)
    return core.device if core is not None else images.device


def _zinfo(ops, Z):
    """(images, patches per image, device) of a rank's embeddings: Z fp32 [n, P, D], or the Taps a lean run keeps instead."""
    if isinstance(Z, Taps):
        return Z.n, ops.zshape()[0], Z.device
    return Z.shape[0], Z.shape[1], Z.device


def _alpha_pair(ops, w_loc, Z_loc, tau, k):
    """(alpha fp64-or-None, alpha fp32 [n_loc, P]) for one tau; w_loc None: the `average` weights (main.py:290-291)."""
    if w_loc is None:
        n, P, device = _zinfo(ops, Z_loc)
        if hasattr(ops, "alpha_average"):
            return None, ops.alpha_average(n, P, device)
        return None, torch.ones(n, P, device=device) / P
    res = ops.alpha_softmax(w_loc, tau, k)
    if isinstance(res, tuple):                                             # the kernel wrote main.py:294's `.float()` too
        return res
    return res, res.float().contiguous()


def _tail(ops, w_loc, Z_loc, counts, group, rank, tau, k, want_pdist, want_host, mark, rot=0):
    """alpha = softmax(w / tau), X = alpha . Z, the all-gathers north_star names, pdist on rank 0 - per tau.
    w does not depend on tau (and Z not on the mode): a tau sweep - the reference re-runs everything per tau
    (examples/main.py:346-366, README "tau_list") - only repeats this cheap tail."""
    taus = list(tau) if isinstance(tau, (list, tuple)) else [tau]
    by_tau = {}
    for tv in taus:
        t0 = time.perf_counter()
        _, alpha_f32 = _alpha_pair(ops, w_loc, Z_loc, tv, k)
        X_loc = ops.weighted_avg_taps(alpha_f32, Z_loc) if isinstance(Z_loc, Taps) else ops.weighted_avg(alpha_f32, Z_loc)
        X = all_gather_rows(X_loc, counts, group, rot)
        alpha = all_gather_rows(alpha_f32, counts, group, rot).unsqueeze(1)
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
    out["by_tau"] = by_tau
    return out


def _marker(ops, timings):
    def mark(name, t0):
        if timings is not None:
            ops.sync()
            timings[name] = timings.get(name, 0.0) + (time.perf_counter() - t0)
    return mark


def run_category(ops, images, supervised="unsupervised", tau=1.0, k=1, train_images=None, train_ratio=1.0,
                 group=None, bank_seed=0, want_pdist=True, want_host=True, timings=None, budget=MINBUF_BYTES,
                 schedule="auto", Z_loc=None, remote_bytes=REMOTE_BYTES, prune=True, solo=False, rot=0, lean=None):
    """images: [N, 3, H, W] (every rank passes the same N; only the rank's shard is read).
    `tau` may be a list: the embeddings and the patch distances are computed once and every tau gets its
    own (alpha, X) in out["by_tau"][tau]; the top-level keys are those of the first tau.
    budget / schedule / remote_bytes: memory budget of one matrix of minima, the unsupervised schedule ("auto" =
    "ring", or "slices") and the memory the ring may spend on receive buffers - results do not depend on any.
    prune: supervised mode as a pruned exact nearest-neighbour search (default) or as exhaustive one-way blocks.
    Z_loc: the rank's embeddings from an earlier call on the same images (out["Z_loc"]): Z does not depend on
    the mode, so a run over several modes embeds the test set once (the reference re-embeds per mode and per tau,
    examples/main.py:346-366).
    solo: work the whole category on THIS rank although a process group exists (no collective at all).
    rot: the shard this rank holds is that of VIRTUAL rank (rank + rot) % world (every rank passes the same rot).
    shard_bounds gives the first ranks ceil(N / world) images and the last one the remainder; a job over many categories
    rotates who is "last" (run_categories) so that no rank holds the large shard of EVERY category.  out["bounds"] is
    indexed by virtual rank; the gathered alpha / X are in image order whatever rot is.
    lean: keep the tapped feature maps (4.8 MB per image) instead of Z (12.85 MB) next to the packed copy (12.85 MB):
    the packed blocks and, after alpha, X are built from the taps LEAN_CHUNK images at a time - the same kernels on the
    same bytes, so nothing in the result changes; out["Z_loc"] is then the Taps (accepted back as `Z_loc`).  None: the
    seam decides from the free HBM (HipOps.wants_lean: N = 10 000 on one GPU is 257 GB with Z, 177 GB without).
    Returns dict(alpha fp32 [N, 1, P] on device, X float32 ndarray [N, D], X_dev, pdist (rank 0), bounds)."""
    real_rank, world = (0, 1) if solo else _world(group)
    rot = rot % world
    rank = (real_rank + rot) % world                                   # virtual rank: indexes bounds, drives the schedule
    n = images.shape[0]
    bounds = shard_bounds(n, world)
    lo, hi = bounds[rank]
    counts = [b - a for a, b in bounds]
    mark = _marker(ops, timings)

    # several ranks: the ring carries the tapped feature maps (4.8 MB per image) instead of packed embeddings (12.85 MB)
    # whenever the seam can turn received taps into packed blocks; the unsupervised "slices" schedule gathers packed Z
    use_taps = world > 1 and getattr(ops, "ring_payload", None) == "taps" and not (supervised == "unsupervised" and schedule == "slices")
    if isinstance(Z_loc, Taps):
        lean = True
    elif Z_loc is not None:
        lean = False
    elif lean is None:
        n_bank = 0 if train_images is None or supervised != "supervised" else -(-min(n, train_images.shape[0]) // world)
        lean = bool(getattr(ops, "wants_lean", lambda _n: False)((hi - lo) + n_bank))
    taps_loc = None
    t0 = time.perf_counter()
    if Z_loc is None:
        if lean:
            Z_loc = taps_loc = ops.embed_taps(images[lo:hi])           # Z is never held as a whole
        elif use_taps and supervised == "unsupervised":
            Z_loc, taps_loc = ops.embed(images[lo:hi], want_taps=True)
        else:
            Z_loc = ops.embed(images[lo:hi])                           # [n_loc, P, D]
    else:
        if _zinfo(ops, Z_loc)[0] != hi - lo:
            raise ValueError("Z_loc holds %d images, this rank's shard has %d" % (_zinfo(ops, Z_loc)[0], hi - lo))
        if lean:
            taps_loc = Z_loc
    mark("embed", t0)
    device = Z_loc.device

    t0 = time.perf_counter()
    if supervised == "supervised":
        if train_images is None:
            raise ValueError("supervised alpha needs train_images")
        sel = select_bank(train_images.shape[0], n, train_ratio, bank_seed)
        bb = shard_bounds(len(sel), world)
        blo, bhi = bb[rank]
        bank_taps = None
        if lean:
            bank_taps = ops.embed_taps(train_images[sel[blo:bhi]])
            pk_bank = ops.pack_taps(bank_taps)
        elif use_taps:
            bank_loc, bank_taps = ops.embed(train_images[sel[blo:bhi]], want_taps=True)
            pk_bank = ops.pack(bank_loc)
            del bank_loc
        else:
            pk_bank = ops.pack(ops.embed(train_images[sel[blo:bhi]]))
        w_loc = alpha_weights_supervised(ops, ops.pack_taps(taps_loc) if lean else ops.pack(Z_loc), pk_bank, bb, rank, world,
                                         group, device, budget, remote_bytes, prune, send=bank_taps if use_taps else None, rot=rot)
        del pk_bank, bank_taps
    elif supervised == "unsupervised":
        w_loc = alpha_weights_unsupervised(ops, ops.pack_taps(taps_loc) if lean else ops.pack(Z_loc), bounds, rank, world, group,
                                           device, budget, schedule, remote_bytes, send=taps_loc if use_taps else None, rot=rot)
        taps_loc = None
    else:                                                              # "average" (main.py:290-291)
        w_loc = None
    mark("pair_min_dist", t0)

    out = _tail(ops, w_loc, Z_loc, counts, group, real_rank, tau, k, want_pdist, want_host, mark, rot)
    out["bounds"] = bounds
    out["rot"] = rot
    out["w_loc"] = w_loc
    out["Z_loc"] = Z_loc
    return out


# ------------------------------------------------------------------ several categories as one job
GROUP_IMAGES = 2048           # images (test + bank) one rank embeds and packs before their alpha work starts: 2 x 26 GB
RING_BYTES_ALL = 48 << 30     # receive buffers of ALL rings that are in flight at once


def lpt_assignment(costs, world):
    """Longest-processing-time-first: item indices per rank, heaviest first onto the least loaded rank (ties: the
    lower rank).  Deterministic, so every rank computes the same table.  -> (list of index lists, loads)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads, mine = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (loads[q], q))
        mine[r].append(i)
        loads[r] += costs[i]
    return [sorted(m) for m in mine], loads


def category_cost(n, m=None, t_img=0.46e-3, t_pair=6.2e-6):
    """Seconds one MI355X spends on a category of n test images (m bank images: supervised): n ViT forwards + the
    ordered alpha pairs.  Defaults: measured on MI355X (0.46 ms per image at batch 128;
    42.2 ms for the 83 * 82 ordered pairs of config 2; a pruned bank pair costs 0.36 of an exhaustive unsupervised one)."""
    if m is None:
        return n * t_img + n * (n - 1) * t_pair
    return (n + m) * t_img + n * m * t_pair * 0.72


def category_rot(c, world):
    """The rotation run_categories gives category number c ("pipeline" schedule): rank r holds the shard of virtual rank
    (r + c) % world, so the short last shard of shard_bounds - and the ceil(N / world) ones - move round the ranks from
    category to category (15 MVTec categories on 8 ranks: 212-219 images per rank instead of 221 on seven ranks and 178
    on the last; 1 % off the slowest rank, free)."""
    return c % world if world > 1 else 0


def _embed_sets(ops, sets, want_taps=False):
    """[Z per set] (+ [Taps per set] when want_taps): one stream of ViT batches when the seam has embed_many."""
    if hasattr(ops, "embed_many"):
        return ops.embed_many(sets, want_taps=True) if want_taps else ops.embed_many(sets)
    if want_taps:
        pairs = [ops.embed(s, want_taps=True) for s in sets]
        return [p[0] for p in pairs], [p[1] for p in pairs]
    return [ops.embed(s) for s in sets]


def _groups(sizes, limit):
    """Consecutive runs of categories whose per-rank image counts stay within `limit` (a single larger one alone)."""
    out, cur, tot = [], [], 0
    for i, sz in enumerate(sizes):
        if cur and tot + sz > limit:
            out.append(cur)
            cur, tot = [], 0
        cur.append(i)
        tot += sz
    if cur:
        out.append(cur)
    return out


def run_categories(ops, items, supervised="unsupervised", tau=1.0, k=1, train_ratio=1.0, group=None, bank_seed=0,
                   want_pdist=True, want_host=True, timings=None, budget=MINBUF_BYTES, remote_bytes=REMOTE_BYTES,
                   prune=True, schedule="pipeline", group_images=GROUP_IMAGES, ring_bytes_all=RING_BYTES_ALL,
                   costs=None, solo=False, Z_locs=None, rotate=True, lean=None):
    """Several categories as ONE job.  items: [{"images": [N_c, 3, H, W], "train_images": bank candidates or None}].
    Returns [run_category's dict per category]; the numbers are those of the per-category loop the reference runs
    (examples/main.py:346-366) - bit for bit given the same embeddings - only the ORDER of the work changes:

      "pipeline" (default)  every category sharded over all ranks, as run_category does, but
           1. the rank's shards of ALL categories (and of their banks) go through the ViT as one stream of FULL batches
              (8 ranks: 5-21 images per category and rank, 216 over the 15 MVTec categories);
           2. every category is packed and the transfers of ALL rings are posted, category by category, before any
              alpha work: the shards of category c + 1 travel while the blocks of category c are multiplied;
           3. the ring step that needs no remote data (the rank's own shard: symmetric block / own bank shard) runs for
              ALL categories first - tens of ms of MFMA work for the first transfers to hide under - then the remaining
              steps, category by category;
           4. the fixed-point scales of all categories come from ONE MAX all-reduce (a vector), their int64 sums from
              ONE SUM all-reduce (one buffer), the supervised overflow flags from ONE MAX all-reduce and ONE host read.
      "sequential"   run_category per category, one after another (the reference's loop; the round-3 behaviour).
      "per_gpu"      whole categories to ranks by longest-processing-time-first on category_cost (or `costs`); every
           rank works its categories alone (no ring, no all-reduce) and ONE all-gather hands X / alpha of every category
           to every rank.  Bound: sum(cost) / max over ranks - 15 MVTec categories on 8 GPUs: 6.4x; SURVEY 8e lists it
           as the alternative to the image-sharded schedule north_star specifies.

    Categories are taken in groups of at most `group_images` images per rank (Z + packed copy: 25.7 MB per image).
    solo: this rank works ALL of `items` alone although a process group exists (how "per_gpu" runs a rank's share).
    Z_locs: the rank's test-set embeddings of EVERY category from an earlier call (out[c]["Z_loc"]; "pipeline" and
    "sequential" only): a second mode on the same images embeds only its bank.
    rotate ("pipeline"): category c is sharded with rotation category_rot(c, world) (see run_category's `rot`): a caller
    that loads only its own shard of the images must load that of virtual rank (rank + category_rot(c, world)) % world.
    lean ("pipeline"): None - every group decides from the rank's free HBM, the ranks agreeing by one MAX all-reduce and one
    host read per group unless the group is too small to matter (HipOps.never_lean); False / True - the caller has decided
    (every rank must pass the same value): no query, no collective, no host read (bench.py decides once, before its
    timed steps)."""
    rank, world = (0, 1) if solo else _world(group)
    mark = _marker(ops, timings)
    ncat = len(items)
    if ncat == 0:
        return []
    if supervised == "supervised" and any(it.get("train_images") is None for it in items):
        raise ValueError("supervised alpha needs train_images for every category")

    if Z_locs is not None and (len(Z_locs) != ncat or any(z is None for z in Z_locs)):
        raise ValueError("Z_locs must hold the embeddings of every category")

    def one(c, rot=0):
        it = items[c]
        return run_category(ops, it["images"], supervised, tau=tau, k=k, train_images=it.get("train_images"),
                            Z_loc=Z_locs[c] if Z_locs is not None else None,
                            train_ratio=train_ratio, group=group, bank_seed=bank_seed, want_pdist=want_pdist,
                            want_host=want_host, timings=timings, budget=budget, remote_bytes=remote_bytes, prune=prune,
                            solo=solo, rot=rot)

    if schedule == "sequential":
        return [one(c) for c in range(ncat)]
    if schedule == "per_gpu":
        return _run_categories_per_gpu(ops, items, supervised, tau, k, train_ratio, group, bank_seed, want_pdist,
                                       want_host, timings, budget, prune, costs, group_images)
    if schedule != "pipeline":
        raise ValueError("unknown category schedule %r" % (schedule,))

    ns = [int(it["images"].shape[0]) for it in items]
    all_bounds = [shard_bounds(n, world) for n in ns]
    rots = [category_rot(c, world) if rotate else 0 for c in range(ncat)]
    vr = [(rank + rots[c]) % world for c in range(ncat)]               # this rank's virtual rank per category
    sels, bank_bounds = [None] * ncat, [None] * ncat
    if supervised == "supervised":
        for c, it in enumerate(items):
            sels[c] = select_bank(it["train_images"].shape[0], ns[c], train_ratio, bank_seed)
            bank_bounds[c] = shard_bounds(len(sels[c]), world)
    # per-rank load of a category: the LARGEST shard (every rank must cut the same groups)
    per_rank = [max(b - a for a, b in all_bounds[c]) + (max(b - a for a, b in bank_bounds[c]) if bank_bounds[c] else 0)
                for c in range(ncat)]
    outs = [None] * ncat
    for grp in _groups(per_rank, group_images):
        # A group that must run LEAN (taps instead of Z: the rank's Z + packed copy of the group would not fit, or the
        # caller hands back the Taps an earlier lean run returned as out["Z_loc"]) goes category by category through
        # run_category, which has the lean path; the one-job order below holds full Z.  Every rank must take the same
        # branch (the two orders post different collectives), so the ranks agree on the flag first.  (Round-4 advice: the
        # one-job path used to read Z_locs[c].shape of a Taps object and to embed large groups with full Z.)
        has_taps = Z_locs is not None and any(isinstance(Z_locs[c], Taps) for c in grp)      # may differ from rank to rank
        grp_images = sum(per_rank[c] for c in grp)                                           # test + bank images of the largest shards
        small = bool(getattr(ops, "never_lean", lambda _n: False)(grp_images))               # shapes only: the same on every rank
        if lean is not None:
            lean_grp = has_taps or bool(lean)          # the caller's word (and its Z_locs) must be the same on every rank
        elif small and Z_locs is None:
            lean_grp = False                           # nobody can want it, nobody can hold Taps: nothing to agree on
        else:
            lean_grp = has_taps or (not small and bool(getattr(ops, "wants_lean", lambda _n: False)(grp_images)))
            if world > 1:
                flag = torch.tensor([int(lean_grp)], dtype=torch.int32, device=_flag_device(ops, items[grp[0]]["images"]))
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
                lean_grp = bool(int(flag.item()))
        if lean_grp:
            for c in grp:
                outs[c] = one(c, rots[c])
            continue
        # ---- 1. one stream of ViT batches for the group
        t0 = time.perf_counter()
        sets = []
        if Z_locs is None:
            sets = [items[c]["images"][all_bounds[c][vr[c]][0]: all_bounds[c][vr[c]][1]] for c in grp]
        else:
            for c in grp:
                if _zinfo(ops, Z_locs[c])[0] != all_bounds[c][vr[c]][1] - all_bounds[c][vr[c]][0]:
                    raise ValueError("Z_locs[%d] holds %d images, this rank's shard has %d"
                                     % (c, _zinfo(ops, Z_locs[c])[0], all_bounds[c][vr[c]][1] - all_bounds[c][vr[c]][0]))
        ntest = len(sets)
        if supervised == "supervised":
            for c in grp:
                blo, bhi = bank_bounds[c][vr[c]]
                sets.append(items[c]["train_images"][sels[c][blo:bhi]])
        # several ranks: the rings carry the tapped feature maps of the shards (see run_category)
        use_taps = world > 1 and getattr(ops, "ring_payload", None) == "taps" and supervised in ("unsupervised", "supervised")
        taps = None
        if use_taps and sets:
            Zs, taps = _embed_sets(ops, sets, want_taps=True)
        else:
            Zs = _embed_sets(ops, sets) if sets else []
        Z = {c: (Zs[i] if Z_locs is None else Z_locs[c]) for i, c in enumerate(grp)}
        Zb = {c: Zs[ntest + i] for i, c in enumerate(grp)} if supervised == "supervised" else {}
        # what each category's ring sends: unsupervised - the taps of its test shard; supervised - of its bank shard
        send = {c: None for c in grp}
        if taps is not None:
            if supervised == "supervised":
                send = {c: taps[ntest + i] for i, c in enumerate(grp)}
            elif Z_locs is None:
                send = {c: taps[i] for i, c in enumerate(grp)}
        del Zs, taps
        mark("embed", t0)
        device = Z[grp[0]].device

        # ---- 2.-4. the alpha weights of the group
        t0 = time.perf_counter()
        w = {c: None for c in grp}
        if supervised in ("unsupervised", "supervised"):
            pk = {c: ops.pack(Z[c]) for c in grp}
            P = pk[grp[0]].P
            sup = supervised == "supervised"
            if sup:
                pkb = {c: ops.pack(Zb[c]) for c in grp}
                Zb.clear()
            if sup and not _can_prune(ops, pk[grp[0]], prune):
                for c in grp:          # exhaustive blocks: no phases to interleave
                    w[c] = alpha_weights_supervised(ops, pk[c], pkb[c], bank_bounds[c], vr[c], world, group, device, budget,
                                                    remote_bytes, prune, send=send[c], rot=rots[c])
            else:
                # the rings of the group must ALL be fully posted (depth = every step); a group whose receive buffers
                # exceed the budget falls back to the category-by-category order (each ring double-buffers on its own)
                steps = (world - 1) if sup else world // 2

                def ring_bytes(c):          # receive buffers of category c's ring with every step posted
                    per_image = send[c].bytes_per_image() if send[c] is not None else (pk[c].D * 4 + 4) * P
                    return steps * per_image * max((b - a) for a, b in (bank_bounds[c] if sup else all_bounds[c]))
                together = world > 1 and sum(ring_bytes(c) for c in grp) <= ring_bytes_all and all(
                    ring_bytes(c) <= remote_bytes for c in grp)
                m_all = ops.new_acc(len(grp), MEAN, device).view(torch.float32)       # 2 fp32 maxima per category, zeros
                jobs = {}
                if sup:
                    flags = ops.new_acc(len(grp), MEAN, device)                          # int64 zeros; int32 view below
                    flags32 = flags.view(torch.int32)[: len(grp)]
                    for i, c in enumerate(grp):
                        jobs[c] = _SupJob(ops, pk[c], pkb[c], bank_bounds[c], vr[c], world, group, device, budget, remote_bytes,
                                          m_out=m_all[2 * i: 2 * i + 2], overflow=flags32[i: i + 1], send=send[c], rot=rots[c])
                else:
                    acc_all = ops.new_acc(sum(ns[c] for c in grp) * P, MEAN, device)
                    off = 0
                    for i, c in enumerate(grp):
                        jobs[c] = _UnsupJob(ops, pk[c], all_bounds[c], vr[c], world, group, device, budget, remote_bytes,
                                            m_out=m_all[2 * i: 2 * i + 1], acc=acc_all[off * P: (off + ns[c]) * P],
                                            send=send[c], rot=rots[c])
                        off += ns[c]
                if world > 1:
                    dist.all_reduce(m_all, op=dist.ReduceOp.MAX, group=group)
                for c in grp:
                    jobs[c].prepare()
                if together or world == 1:
                    for c in grp:
                        jobs[c].post()
                    for c in grp:                      # no remote data needed: the first transfers hide under ALL of these
                        jobs[c].run(0)
                    for c in grp:
                        jobs[c].run_later()
                else:
                    for c in grp:
                        jobs[c].post()
                        jobs[c].run(0)
                        jobs[c].run_later()
                        jobs[c].ring = None
                if sup:
                    if world > 1:
                        dist.all_reduce(flags32, op=dist.ReduceOp.MAX, group=group)
                    over = [bool(v) for v in flags32.cpu().tolist()]                     # ONE host read for the group
                    for i, c in enumerate(grp):
                        if over[i]:
                            jobs[c].fix()
                elif world > 1:
                    dist.all_reduce(acc_all, op=dist.ReduceOp.SUM, group=group)
                for c in grp:
                    w[c] = jobs[c].finish()
                del jobs
            del pk
        mark("pair_min_dist", t0)

        # ---- the cheap tail, per category
        for c in grp:
            counts = [b - a for a, b in all_bounds[c]]
            out = _tail(ops, w[c], Z[c], counts, group, rank, tau, k, want_pdist, want_host, mark, rots[c])
            out["bounds"] = all_bounds[c]
            out["rot"] = rots[c]
            out["w_loc"] = w[c]
            out["Z_loc"] = Z[c]
            outs[c] = out
        del Z
    return outs


def _run_categories_per_gpu(ops, items, supervised, tau, k, train_ratio, group, bank_seed, want_pdist, want_host, timings,
                            budget, prune, costs, group_images):
    """Whole categories to ranks (see run_categories): every rank runs ITS categories as a one-rank pipeline job, then ONE
    all-gather of rows (X | alpha) per tau puts every category's result on every rank; rank 0 computes the distances."""
    rank, world = _world(group)
    mark = _marker(ops, timings)
    ns = [int(it["images"].shape[0]) for it in items]
    if costs is None:
        costs = [category_cost(n, min(int(it["train_images"].shape[0]), int(train_ratio * n)) if supervised == "supervised" else None)
                 for n, it in zip(ns, items)]
    table, _ = lpt_assignment(list(costs), world)
    mine = table[rank]
    local = run_categories(ops, [items[c] for c in mine], supervised, tau=tau, k=k, train_ratio=train_ratio, group=group,
                           bank_seed=bank_seed, want_pdist=False, want_host=False, timings=timings, budget=budget,
                           prune=prune, schedule="pipeline", group_images=group_images, solo=True) if mine else []
    taus = list(tau) if isinstance(tau, (list, tuple)) else [tau]
    outs = [dict(bounds=shard_bounds(n, 1), w_loc=None, Z_loc=None, by_tau={}) for n in ns]
    for r, cats in enumerate(table):
        for c in cats:
            outs[c]["owner"] = r
    for c, res in zip(mine, local):
        outs[c]["w_loc"], outs[c]["Z_loc"] = res["w_loc"], res["Z_loc"]
    dims = (int(local[0]["X_dev"].shape[1]), int(local[0]["alpha"].shape[2])) if local else None
    if world > 1 and any(not cats for cats in table):          # more ranks than categories: an idle rank asks for the row width
        seen = [None] * world
        dist.all_gather_object(seen, dims, group=group)
        dims = next(v for v in seen if v is not None)
    rows = [sum(ns[c] for c in cats) for cats in table]
    device = local[0]["X_dev"].device if local else getattr(items[0]["images"], "device", None)
    for tv in taus:
        t0 = time.perf_counter()
        full = None
        if world > 1:
            D, P = dims
            send = torch.empty((rows[rank], D + P), dtype=torch.float32, device=device)
            off = 0
            for c, res in zip(mine, local):
                r_tv = res["by_tau"][float(tv)]
                send[off: off + ns[c], :D].copy_(r_tv["X_dev"])
                send[off: off + ns[c], D:].copy_(r_tv["alpha"].squeeze(1))
                off += ns[c]
            full = all_gather_rows(send, rows, group)
        off = 0
        for r, cats in enumerate(table):
            for c in cats:
                if full is not None:
                    blk = full[off: off + ns[c]]
                    off += ns[c]
                    res = {"X_dev": blk[:, :dims[0]].contiguous(), "alpha": blk[:, dims[0]:].contiguous().unsqueeze(1)}
                else:
                    src = local[mine.index(c)]["by_tau"][float(tv)]
                    res = {"X_dev": src["X_dev"], "alpha": src["alpha"]}
                outs[c]["by_tau"][float(tv)] = res
        mark("alpha_softmax_weighted_avg_gather", t0)
        t0 = time.perf_counter()
        for c in range(len(items)):
            res = outs[c]["by_tau"][float(tv)]
            if want_pdist and rank == 0:
                res["pdist"] = ops.pdist(res["X_dev"])
            if want_host:
                res["X"] = np.array(res["X_dev"].cpu().numpy(), dtype=np.float32, copy=True)
        mark("pdist", t0)
    for c in range(len(items)):
        outs[c].update(outs[c]["by_tau"][float(taus[0])])
    return outs
