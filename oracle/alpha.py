"""Oracle: per-image patch weights alpha and the weighted-average embedding.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Restates models/patchcore/utils.py:222-277 and examples/main.py:286-296.
The reference grows tensors with torch.cat inside Python loops; the
restatement keeps the arithmetic (per ordered image pair: fp32 torch.cdist ->
row-min; then mean / min over images; fp64 exp/normalise) and drops the
O(N^2) reallocation.
"""
import math

import numpy as np
import torch


def pair_min_distance(zi, zj):
    """utils.py:226 / :233-234: min over patches q of ||zi[p] - zj[q]||_2.
    [P, D] x [Q, D] -> [P] fp32 (torch.cdist, p=2, default compute mode)."""
    return torch.min(torch.cdist(zi, zj), dim=1)[0]


@torch.no_grad()
def weight_distance_unsupervised(Z, i):
    """utils.py:222-227: w_i[p] = mean_{j != i} min_q ||Z_i[p] - Z_j[q]||."""
    cols = [pair_min_distance(Z[i], Z[j]).unsqueeze(1) for j in range(Z.shape[0]) if j != i]
    return torch.mean(torch.cat(cols, dim=1), dim=1)


@torch.no_grad()
def weight_distance_supervised(Z, Z_train, i):
    """utils.py:230-237: w_i[p] = min_j min_q ||Z_i[p] - Ztrain_j[q]||."""
    cols = [pair_min_distance(Z[i], Z_train[j]).unsqueeze(1) for j in range(Z_train.shape[0])]
    return torch.min(torch.cat(cols, dim=1), dim=1)[0]


def alpha_from_weights(w, tau, k=1):
    """utils.py:246-254 / :266-274 for one image. w: [P] fp32 -> [1, P] fp64.
    tau ~ 0 -> indicator of the maximum normalised over ties; otherwise the
    un-shifted k*exp(w/tau)/sum in float64."""
    w = w.unsqueeze(0).double()
    if math.isclose(tau, 0):
        a = (w == w.max(axis=1, keepdims=True)[0]).to(dtype=torch.float64)
    else:
        a = k * torch.exp((1 / tau) * w)
    return a / a.sum()


@torch.no_grad()
def weights_unsupervised(Z):
    """All w_i for the unsupervised mode: [N, P] fp32."""
    return torch.stack([weight_distance_unsupervised(Z, i) for i in range(Z.shape[0])])


@torch.no_grad()
def weights_supervised(Z, Z_train):
    return torch.stack([weight_distance_supervised(Z, Z_train, i) for i in range(Z.shape[0])])


def matrix_alpha_unsupervised(tau, k, Z):
    """utils.py:240-257 -> DoubleTensor [N, P]."""
    return torch.cat([alpha_from_weights(w, tau, k) for w in weights_unsupervised(Z)], dim=0)


def matrix_alpha_supervised(tau, k, Z, Z_train):
    """utils.py:260-277 -> DoubleTensor [N, P]."""
    return torch.cat([alpha_from_weights(w, tau, k) for w in weights_supervised(Z, Z_train)], dim=0)


def matrix_alpha_average(Z):
    """examples/main.py:290-291."""
    return torch.ones(Z.shape[0], Z.shape[1]) / Z.shape[1]


@torch.no_grad()
def weighted_average(matrix_alpha, Z):
    """examples/main.py:294-296: alpha -> [N,1,P] float32, X = bmm(alpha, Z) as
    np.float32 [N, D].  Returns (alpha [N,1,P] fp32, X)."""
    a = matrix_alpha.unsqueeze(1).float()
    X = torch.bmm(a, Z).squeeze(1).cpu().numpy().copy()
    return a, X


FLOP_PER_PAIR = lambda P, D: 2.0 * P * P * D  # 5.035 GFLOP at P=784, D=4096
