"""AnomalyClusteringCore / PatchMaker on the MI355X kernels
(reference: models/patchcore/patchcore.py:277-479).

Same surface as the reference: ``AnomalyClusteringCore(device).load(backbone, layers_to_extract_from,
device, input_shape, pretrain_embed_dimension, target_embed_dimension, patchsize=3, patchstride=1,
anomaly_score_num_nn=1, featuresampler=IdentitySampler(), nn_method=FaissNN(False, 4), **kwargs)``,
``.embed(dataloader_or_tensor, supervised)``, ``._embed(images, supervised, detach=True,
provide_patch_shapes=False)`` and the attributes ``forward_modules[...]``, ``patch_maker``,
``backbone``, ``layers_to_extract_from``.

Differences, all deliberate:
  * `_embed` after the backbone is ONE fused HIP kernel (CLS drop, per-image standardisation,
    zero-padded unfold, grid alignment, both adaptive pools): the 9x unfolded tensor never
    exists in HBM (reference moves >100 MB/image for 17.7 MB of essential traffic).
  * `embed_device(...)` is added: batched, device-resident `[N, P, D]` output.  `embed(...)`
    keeps the reference's return type (list per image of 784 numpy rows) for drop-in callers,
    at the price of the D2H copy the reference also pays.
  * the reference's fresh LayerNorm leaves `requires_grad` on its output with detach=False;
    a plain tensor is returned here.
  * `PatchCore` (anomaly detection: fit / predict / save / load, :20-275) is out of scope.
"""
import math

import numpy as np
import torch
import tqdm

import acnative
import patchcore.common
import patchcore.sampler


class PatchMaker:
    def __init__(self, patchsize, stride=None):
        self.patchsize = patchsize
        self.stride = stride

    def patch_grid(self, height, width):
        """Number of patches per axis after the zero-padded unfold (reference :447-457)."""
        padding = int((self.patchsize - 1) / 2)
        stride = self.stride or 1
        return [int((s + 2 * padding - 1 * (self.patchsize - 1) - 1) / stride + 1) for s in (height, width)]

    def patchify(self, features, return_spatial_info=False):
        """[B, C, H, W] -> [B, H'*W', C, k, k] (reference :439-465).  Materialises the unfold on
        the device with a gather kernel; the fused `_embed` path never calls this."""
        unfolded = acnative.unfold_patches(features, self.patchsize, self.stride or 1)
        if return_spatial_info:
            return unfolded, self.patch_grid(*features.shape[-2:])
        return unfolded

    def unpatch_scores(self, x, batchsize):
        return x.reshape(batchsize, -1, *x.shape[1:])

    def score(self, x):
        was_numpy = isinstance(x, np.ndarray)
        if was_numpy:
            x = torch.from_numpy(x)
        while x.ndim > 1:
            x = torch.max(x, dim=-1).values
        return x.numpy() if was_numpy else x


def _batch_edges(n, batch_size):
    """[(first, last)] of the ViT launches over n images: `batch_size` images each, or - when it is a list - exactly those
    sizes in turn (pipeline.vit_batch_plan: launch sizes whose tile grids fill whole rounds of the chip)."""
    if isinstance(batch_size, (list, tuple)):
        if sum(batch_size) != n or any(b <= 0 for b in batch_size):
            raise ValueError("batch plan %r does not cover %d images" % (list(batch_size), n))
        edges, s0 = [], 0
        for b in batch_size:
            edges.append((s0, s0 + b))
            s0 += b
        return edges
    return [(s0, min(n, s0 + batch_size)) for s0 in range(0, n, batch_size)]


class AnomalyClusteringCore(torch.nn.Module):
    def __init__(self, device):
        super().__init__()
        self.device = device

    def load(self, backbone, layers_to_extract_from, device, input_shape, pretrain_embed_dimension,
             target_embed_dimension, patchsize=3, patchstride=1, anomaly_score_num_nn=1,
             featuresampler=patchcore.sampler.IdentitySampler(), nn_method=None, **kwargs):
        self.backbone = backbone.to(device)
        self.layers_to_extract_from = layers_to_extract_from
        self.input_shape = input_shape
        self.device = device
        self.patch_maker = PatchMaker(patchsize, stride=patchstride)
        self.forward_modules = torch.nn.ModuleDict({})
        feature_aggregator = patchcore.common.NetworkFeatureAggregator(
            self.backbone, self.layers_to_extract_from, self.device)
        feature_dimensions = feature_aggregator.feature_dimensions(input_shape)
        self.forward_modules["feature_aggregator"] = feature_aggregator
        self.forward_modules["preprocessing"] = patchcore.common.Preprocessing(
            feature_dimensions, pretrain_embed_dimension)
        self.target_embed_dimension = target_embed_dimension
        preadapt_aggregator = patchcore.common.Aggregator(target_dim=target_embed_dimension)
        _ = preadapt_aggregator.to(self.device)
        self.forward_modules["preadapt_aggregator"] = preadapt_aggregator
        self.anomaly_scorer = patchcore.common.NearestNeighbourScorer(
            n_nearest_neighbours=anomaly_score_num_nn,
            nn_method=nn_method if nn_method is not None else patchcore.common.FaissNN(False, 4))
        self.anomaly_segmentor = patchcore.common.RescaleSegmentor(
            device=self.device, target_size=input_shape[-2:])
        self.featuresampler = featuresampler
        self._zdims = None           # (P, D) of one image's Z, known after the first embedding

    # ------------------------------------------------------------------ embedding
    def embed(self, data, supervised):
        """Reference :337-353.  DataLoader of dict batches -> (features, labels) with
        features[i] = list of P numpy rows, labels[i] = the batch's `is_anomaly`."""
        print("{:-^80}".format("embedding"))
        if isinstance(data, torch.utils.data.DataLoader):
            features, labels = [], []
            with tqdm.tqdm(total=len(data)) as progress:
                for image in data:
                    if isinstance(image, dict):
                        is_anomaly = image["is_anomaly"]
                        image = image["image"]
                    with torch.no_grad():
                        input_image = image.to(torch.float).to(self.device)
                        features.append(self._embed(input_image, supervised))
                        labels.append(is_anomaly)
                    progress.update(1)
            return features, labels
        return self._embed(data, supervised)

    @torch.no_grad()
    def embed_device(self, images, batch_size=32):
        """Batched, device-resident embedding: images [N, 3, H, W] (any device) -> Z [N, P, D] fp32
        on self.device.  This is what the MI355X pipeline (examples/main.py, bench.py) uses.  Z is allocated
        once and every batch's aggregation kernel writes its rows in place (no concatenation copy)."""
        N = images.shape[0]
        dims = getattr(self, "_zdims", None)
        edges = _batch_edges(N, batch_size)
        Z = torch.empty((N,) + dims, dtype=torch.float32, device=self.device) if (dims and len(edges) > 1) else None
        for s, e in edges:
            chunk = images[s:e].to(self.device, dtype=torch.float32, non_blocking=True)
            if Z is None:
                z = self._embed(chunk, None, detach=False)
                rows = z.shape[0] // chunk.shape[0]
                self._zdims = (rows, z.shape[-1])
                if chunk.shape[0] == N:
                    return z.reshape(N, rows, z.shape[-1])
                Z = torch.empty((N, rows, z.shape[-1]), dtype=torch.float32, device=self.device)
                Z[: chunk.shape[0]] = z.reshape(chunk.shape[0], rows, -1)
            else:
                self._embed(chunk, None, detach=False, out=Z[s:s + chunk.shape[0]])
        return Z

    @torch.no_grad()
    def aggregate_taps(self, feats):
        """The tapped feature maps of some images (as embed_device(..., want_taps=True) returned them, e.g. received
        from another rank) -> Z fp32 [n, P, D]: the same aggregation kernel on the same bytes, so the same Z bit for bit.
        A rank sends 4.8 MB of taps per image instead of 12.85 MB of packed Z and the receiver re-aggregates."""
        n = feats[0].shape[0]
        Zr, _ = acnative.patch_aggregate(list(feats), self.patch_maker.patchsize,
                                         self.forward_modules["preprocessing"].output_dim,
                                         self.forward_modules["preadapt_aggregator"].target_dim)
        return Zr.reshape(n, Zr.shape[0] // n, Zr.shape[-1])

    @torch.no_grad()
    def aggregate_taps_hl32(self, feats, out_hl):
        """aggregate_taps with the rows written straight into the hl32 rows `out_hl` (Packed.hl of the block): the bytes
        pack_images(aggregate_taps(feats)).hl would hold, without the fp32 Z (12.85 MB per image written and read back) in
        between.  For blocks that are only multiplied - the shards received over the ring.  False when the shape is not the
        one the fused kernel serves (the caller then aggregates and packs)."""
        args = (list(feats), self.patch_maker.patchsize, self.forward_modules["preprocessing"].output_dim,
                self.forward_modules["preadapt_aggregator"].target_dim)
        if not acnative.patch_aggregate_hl32_supported(*args):
            return False
        acnative.patch_aggregate_hl32(*args, out_hl)
        return True

    @torch.no_grad()
    def embed_taps_device(self, images, batch_size=32):
        """Only the tapped feature maps of the images ([N, ...] per extracted layer, device-resident), no aggregation:
        what a category too large to hold Z (12.85 MB per image) next to its packed copy keeps instead - 4.8 MB per
        image for ViT-B/8, from which aggregate_taps rebuilds any block of Z bit for bit."""
        N = images.shape[0]
        bufs = None
        _ = self.forward_modules["feature_aggregator"].eval()
        for s, e in _batch_edges(N, batch_size):
            chunk = images[s:e].to(self.device, dtype=torch.float32, non_blocking=True)
            taps = self.forward_modules["feature_aggregator"](chunk)
            feats = [taps[layer] for layer in self.layers_to_extract_from]
            if not all(t.is_contiguous() for t in feats):
                raise RuntimeError("embed_taps_device needs contiguous feature maps (token maps); this backbone's are strided views")
            if chunk.shape[0] == N:
                return feats
            if bufs is None:
                bufs = [torch.empty((N,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in feats]
            for buf, t in zip(bufs, feats):
                buf[s:s + chunk.shape[0]].copy_(t)
        return bufs

    @torch.no_grad()
    def embed_device_many(self, image_sets, batch_size=32, want_taps=False):
        """Several image sets -> [Z_i fp32 [n_i, P, D]] (views of ONE buffer), embedded as one stream of full
        batches: a batch may span set boundaries (its images are then gathered into a staging buffer by
        device-to-device copies; a batch inside one set is read in place).  The reference embeds one image per
        forward (examples/main.py:213, batch_size=1); a rank of an 8-GPU run holds 5-21 images of an MVTec
        category, and a ViT launch on 11 images runs at 78 % of the per-image rate of a launch on 128."""
        counts = [int(s.shape[0]) for s in image_sets]
        total = sum(counts)
        if total == 0:
            raise ValueError("embed_device_many: no images")
        first = next(s for s in image_sets if s.shape[0])
        shape = tuple(first.shape[1:])
        stage = None
        taps_all = None                                  # want_taps: per layer one [total, ...] buffer, filled batch by batch
        # the row shape of Z is known after the first embedding of this core: Z is then allocated up front and every
        # batch writes in place (the very first call copies its first batch once)
        dims = getattr(self, "_zdims", None)
        Z_all = torch.empty((total,) + dims, dtype=torch.float32, device=self.device) if dims else None
        # (set index, first, last) segments of the global image range
        starts = [0]
        for c in counts:
            starts.append(starts[-1] + c)
        edges = _batch_edges(total, batch_size)
        largest = max(e - s for s, e in edges)
        for s, e in edges:
            segs = [(i, max(s, starts[i]) - starts[i], min(e, starts[i + 1]) - starts[i])
                    for i in range(len(counts)) if starts[i] < e and starts[i + 1] > s]
            if len(segs) == 1:
                i, a, b = segs[0]
                chunk = image_sets[i][a:b].to(self.device, dtype=torch.float32, non_blocking=True)
            else:
                if stage is None:
                    stage = torch.empty((largest,) + shape, dtype=torch.float32, device=self.device)
                off = 0
                for i, a, b in segs:
                    stage[off: off + (b - a)].copy_(image_sets[i][a:b], non_blocking=True)
                    off += b - a
                chunk = stage[: e - s]
            got = [] if want_taps else None
            if Z_all is None:
                z = self._embed(chunk, None, detach=False, taps_out=got)
                rows, D = z.shape[0] // chunk.shape[0], z.shape[-1]
                self._zdims = (rows, D)
                if e == total:
                    Z_all = z.reshape(total, rows, D)
                else:
                    Z_all = torch.empty((total, rows, D), dtype=torch.float32, device=self.device)
                    Z_all[: e - s].copy_(z.reshape(e - s, rows, D))
                del z
            else:
                self._embed(chunk, None, detach=False, out=Z_all[s:e], taps_out=got)
            if want_taps:
                if not all(t.is_contiguous() for t in got):
                    raise RuntimeError("want_taps needs contiguous feature maps (token maps); this backbone's are strided views")
                if e - s == total:
                    taps_all = got                                        # one batch: the maps as the backbone wrote them
                else:
                    if taps_all is None:
                        taps_all = [torch.empty((total,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in got]
                    for buf, t in zip(taps_all, got):
                        buf[s:e].copy_(t)
        Zs = [Z_all[starts[i]: starts[i + 1]] for i in range(len(counts))]
        if want_taps:
            return Zs, [[buf[starts[i]: starts[i + 1]] for buf in taps_all] for i in range(len(counts))]
        return Zs

    def _embed(self, images, supervised, detach=True, provide_patch_shapes=False, out=None, taps_out=None):
        """Reference :355-431.  `supervised` is accepted and ignored, as upstream.  out (extension): a
        contiguous fp32 device tensor the rows of Z are written into.  taps_out (extension): a list that receives the
        tapped feature maps of this batch (what the aggregation kernel reads)."""
        _ = self.forward_modules["feature_aggregator"].eval()
        with torch.no_grad():
            taps = self.forward_modules["feature_aggregator"](images)
            feats = [taps[layer] for layer in self.layers_to_extract_from]
            if taps_out is not None:
                taps_out.extend(feats)
            Z, patch_shapes = acnative.patch_aggregate(
                feats, self.patch_maker.patchsize,
                self.forward_modules["preprocessing"].output_dim,
                self.forward_modules["preadapt_aggregator"].target_dim, out=out)
        out = [x.detach().cpu().numpy() for x in Z] if detach else Z
        if provide_patch_shapes:
            return out, patch_shapes
        return out
