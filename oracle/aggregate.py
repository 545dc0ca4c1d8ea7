"""Oracle: tapped feature maps -> per-patch embedding matrix Z.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Restates AnomalyClusteringCore._embed after the backbone call
(models/patchcore/patchcore.py:367-431), PatchMaker.patchify (:439-465),
Preprocessing / MeanMapper (models/patchcore/common.py:145-170) and
Aggregator (common.py:173-183).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def to_chw(feature):
    """patchcore.py:377-383: a [B, T, C] token map drops CLS and becomes
    [B, C, sqrt(T-1), sqrt(T-1)]; a [B, C, H, W] CNN map passes through."""
    if feature.dim() == 3:
        feature = feature[:, 1:, :]
        side = int(math.sqrt(feature.shape[1]))
        feature = feature.reshape(feature.shape[0], side, side, feature.shape[2]).permute(0, 3, 1, 2)
    return feature


def standardise(feature, eps=1e-5):
    """patchcore.py:384-385: a *fresh* LayerNorm([C, H, W]) (weight 1, bias 0,
    eps 1e-5) = per-image standardisation over all C*H*W values (biased var)."""
    return F.layer_norm(feature, feature.shape[1:], None, None, eps)


def patchify(feature, patchsize, stride=1):
    """patchcore.py:439-465: zero-padded k x k unfold.
    [B, C, H, W] -> ([B, H'*W', C, k, k], [H', W'])."""
    pad = int((patchsize - 1) / 2)
    unfolded = F.unfold(feature, kernel_size=patchsize, stride=stride, padding=pad, dilation=1)
    grid = [int((s + 2 * pad - (patchsize - 1) - 1) / stride + 1) for s in feature.shape[-2:]]
    unfolded = unfolded.reshape(*feature.shape[:2], patchsize, patchsize, -1).permute(0, 4, 1, 2, 3)
    return unfolded, grid


def align_to_reference_grid(patches, grid, ref_grid):
    """patchcore.py:398-421: bilinear (align_corners=False) resize of the
    *patch grid* of a later layer onto the first layer's grid."""
    x = patches.reshape(patches.shape[0], grid[0], grid[1], *patches.shape[2:])
    x = x.permute(0, -3, -2, -1, 1, 2)
    base = x.shape
    x = x.reshape(-1, *x.shape[-2:])
    x = F.interpolate(x.unsqueeze(1), size=(ref_grid[0], ref_grid[1]), mode="bilinear", align_corners=False)
    x = x.squeeze(1).reshape(*base[:-2], ref_grid[0], ref_grid[1])
    x = x.permute(0, -2, -1, 1, 2, 3)
    return x.reshape(len(x), -1, *x.shape[-3:])


def mean_map(x, out_dim):
    """common.py:163-170: flatten each patch to one row, adaptive_avg_pool1d."""
    return F.adaptive_avg_pool1d(x.reshape(len(x), 1, -1), out_dim).squeeze(1)


@torch.no_grad()
def aggregate(features, patchsize=3, pretrain_embed_dimension=2048, target_embed_dimension=4096,
              patchstride=1):
    """features: list (layer order) of [B,T,C] or [B,C,H,W] fp32 CPU tensors.
    Returns Z: [B * P, target_embed_dimension] fp32 (row = patch, row-major over
    the first layer's grid, images concatenated) and the per-layer patch grids."""
    feats = [standardise(to_chw(f.to(torch.float32))) for f in features]
    pg = [patchify(f, patchsize, patchstride) for f in feats]
    patches, grids = [p[0] for p in pg], [p[1] for p in pg]
    ref = grids[0]
    for i in range(1, len(patches)):
        patches[i] = align_to_reference_grid(patches[i], grids[i], ref)
    patches = [p.reshape(-1, *p.shape[-3:]) for p in patches]
    # Preprocessing (common.py:145-160): one MeanMapper per layer, stacked on dim 1
    stacked = torch.stack([mean_map(p, pretrain_embed_dimension) for p in patches], dim=1)
    # Aggregator (common.py:173-183)
    z = F.adaptive_avg_pool1d(stacked.reshape(len(stacked), 1, -1), target_embed_dimension)
    return z.reshape(len(z), -1), grids


def adaptive_windows(length, out):
    """Window [start, end) of torch's adaptive_avg_pool1d for every output bin:
    start = floor(i*L/O), end = ceil((i+1)*L/O)  (SURVEY.md 2.1)."""
    i = np.arange(out, dtype=np.int64)
    start = (i * length) // out
    end = -((-(i + 1) * length) // out)
    return start, end
