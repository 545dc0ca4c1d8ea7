"""Weighted-average embeddings -> Ward clusters -> Hungarian relabel -> NMI / ARI / F1
(reference: examples/test.py:109-131 best_map, :176-220 calculate_metrics).

Everything O(N^2) runs on the GPU: ``acnative.pdist`` (condensed float64 Euclidean distances, the quantity
scipy's Ward linkage starts from) and ``acnative.ward_linkage`` (the nearest-neighbour-chain linkage itself, a
restatement of scipy.cluster.hierarchy.linkage(method="ward") - the routine sklearn's AgglomerativeClustering
reaches through ward_tree - that returns a bit-identical linkage matrix; csrc/cluster.hip).  NMI and ARI come
from the contingency matrix on the GPU (``acnative.cluster_scores``).  What stays on the host is O(N) or
O(k^3) glue: the tree cut (sklearn's _hc_cut numbering), the Hungarian relabelling of <= 9 clusters (scipy's
linear_sum_assignment in place of munkres) and the micro-F1 of the relabelled vector.

Round 5: sklearn's LabelEncoder / f1_score left the path (they were 99 % of the 62 ms this chain took for one N = 83
category: input validation, type-of-target probes, first-call imports) - the label encoding is numpy.unique's inverse
(the same sorted-classes numbering) and the micro-averaged F1 of single-label vectors IS the accuracy: summed over the
classes, every misplaced image is one false positive and one false negative, so precision = recall = TP / N and
2PR / (P + R) = TP / N.
"""
from heapq import heappush, heappushpop

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

import acnative


def ward_children(condensed, n):
    """Ward linkage on a condensed float64 distance vector (DEVICE tensor) -> merge table [n-1, 2] (sklearn's
    children_).  Raises on host data: the linkage has no CPU fallback."""
    if n < 2:
        return np.zeros((0, 2), dtype=np.intp)
    Z = acnative.ward_linkage(condensed, n)
    return Z[:, :2].cpu().numpy().astype(np.intp)


def cut_tree(children, n_clusters, n_leaves):
    """Labels from cutting the merge tree into `n_clusters` (same numbering as sklearn's cut: the
    clusters are enumerated in the order of a min-heap of negated node ids)."""
    if n_clusters > n_leaves:
        raise ValueError("Cannot extract more clusters than samples: %d clusters for %d leaves"
                         % (n_clusters, n_leaves))
    if n_leaves == 1:
        return np.zeros(1, dtype=np.intp)
    heap = [-(int(max(children[-1])) + 1)]
    for _ in range(n_clusters - 1):
        left, right = children[-heap[0] - n_leaves]
        heappush(heap, -int(left))
        heappushpop(heap, -int(right))
    labels = np.zeros(n_leaves, dtype=np.intp)
    for cluster_id, neg in enumerate(heap):
        stack = [-neg]
        while stack:
            node = stack.pop()
            if node < n_leaves:
                labels[node] = cluster_id
            else:
                stack.extend(int(c) for c in children[node - n_leaves])
    return labels


def agglomerative_ward(X, n_clusters, device=None):
    """AgglomerativeClustering(n_clusters).fit_predict(X) with the pairwise distances on the GPU.
    X: float32 array / tensor [N, D]."""
    if not isinstance(X, torch.Tensor):
        X = torch.from_numpy(np.ascontiguousarray(np.asarray(X, dtype=np.float32)))
    if device is None and not X.is_cuda and torch.cuda.is_available():
        device = torch.device("cuda:0")
    if device is not None:
        X = X.to(device)
    n = X.shape[0]
    if n == 1:
        return np.zeros(1, dtype=np.intp)
    condensed = acnative.pdist(X.to(torch.float32).contiguous())    # raises on a CPU tensor: no CPU fallback
    return cut_tree(ward_children(condensed, n), n_clusters, n)


def encode_labels(names):
    """sklearn's LabelEncoder().fit_transform: classes sorted, ids in that order (reference :190-191)."""
    if len(names) == 0:
        return np.zeros(0, dtype=int)
    return np.unique(np.asarray(names), return_inverse=True)[1].astype(int)


def micro_f1(label, predict):
    """sklearn's f1_score(average="micro") of two single-label vectors (reference :220) - the accuracy, in the
    arithmetic sklearn 1.1 uses (precision and recall from the summed counts, then 2PR / (P + R))."""
    label, predict = np.asarray(label), np.asarray(predict)
    if label.size == 0:
        return 0.0
    p = float(np.count_nonzero(label == predict)) / float(label.size)      # precision == recall
    return 0.0 if p == 0.0 else 2.0 * p * p / (p + p)


def best_map(L1, L2):
    """Relabel cluster ids L2 with the labels of L1 that maximise the overlap (reference :109-131)."""
    labels1, labels2 = np.unique(L1), np.unique(L2)
    size = max(len(labels1), len(labels2))
    G = np.zeros((size, size))
    for i, a in enumerate(labels1):
        for j, b in enumerate(labels2):
            G[i, j] = np.sum((L2 == b).astype(float) * (L1 == a).astype(float))
    rows, cols = linear_sum_assignment(-G.T)
    pick = cols[np.argsort(rows)]
    out = np.zeros(L2.shape)
    for i, b in enumerate(labels2):
        out[L2 == b] = labels1[pick[i]]
    return out


def metrics_from_labels(label, predict, device=None):
    """Ground-truth ids + cluster ids -> NMI, ARI (contingency matrix on the GPU), micro-F1 of the Hungarian-
    relabelled prediction (reference :197-220).  Returns NMI, ARI, F1, label, predict(relabelled)."""
    label = np.asarray(label).astype(int)
    predict = best_map(label, np.asarray(predict)).astype(int)
    if device is None:       # no GPU: acnative.cluster_scores raises on the host tensors (no CPU fallback)
        device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
    lt = torch.from_numpy(label.astype(np.int32)).to(device)
    # best_map writes ground-truth label VALUES into predict: both vectors live in [0, max label]
    k = int(max(label.max(), predict.max())) + 1
    NMI, ARI, _ = acnative.cluster_scores(lt, torch.from_numpy(predict.astype(np.int32)).to(device), k, k)
    F1 = micro_f1(label, predict)
    return NMI, ARI, F1, label, predict


def cluster_metrics_from_pdist(condensed, anomaly_names):
    """The same chain from distances that are already on the device (bench.py, pipeline output): condensed
    float64 Euclidean distances of ALL images + per-image anomaly strings.  "combined" images are dropped by
    selecting the sub-matrix (distances between the kept images do not change)."""
    keep = [i for i, a in enumerate(anomaly_names) if a != "combined"]
    names = [anomaly_names[i] for i in keep]
    n_all, n = len(anomaly_names), len(keep)
    if n != n_all:
        idx = torch.tensor(keep, device=condensed.device)
        iu = torch.triu_indices(n, n, 1, device=condensed.device)
        a, b = idx[iu[0]], idx[iu[1]]
        condensed = condensed[a * n_all - a * (a + 1) // 2 + (b - a - 1)].contiguous()
    label = encode_labels(names)
    predict = cut_tree(ward_children(condensed, n), len(set(label)), n) if n > 1 else np.zeros(n, dtype=np.intp)
    return metrics_from_labels(label, predict, condensed.device)


def cluster_metrics(X, anomaly_names, device=None):
    """Reference :176-220 from (X, per-image anomaly strings): drop "combined", Ward with
    k = #labels, best_map, NMI / ARI / micro-F1.  Returns NMI, ARI, F1, label, predict."""
    keep = [i for i, a in enumerate(anomaly_names) if a != "combined"]
    names = [anomaly_names[i] for i in keep]
    Xk = np.asarray(X)[keep]
    label = encode_labels(names)
    predict = agglomerative_ward(Xk, len(set(label)), device)
    return metrics_from_labels(label, predict, device)
