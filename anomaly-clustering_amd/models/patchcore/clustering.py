"""Weighted-average embeddings -> Ward clusters -> Hungarian relabel -> NMI / ARI / F1
(reference: examples/test.py:109-131 best_map, :176-220 calculate_metrics).

The N x N distance work runs on the GPU (``acnative.pdist``: condensed float64 Euclidean
distances, the quantity scipy's Ward linkage starts from); the O(N^2) nearest-neighbour-chain
linkage itself stays on the host in scipy, exactly the routine sklearn's
AgglomerativeClustering(linkage="ward") calls (ward_tree -> scipy.cluster.hierarchy.ward), fed
with the precomputed condensed matrix instead of X.  Tree cut, label matching and the three
scores follow the reference (sklearn metrics; scipy's linear_sum_assignment in place of munkres).
"""
from heapq import heappush, heappushpop

import numpy as np
import torch
from scipy.cluster import hierarchy
from scipy.optimize import linear_sum_assignment
from sklearn import metrics
from sklearn.preprocessing import LabelEncoder

import acnative


def ward_children(condensed, n):
    """Ward linkage on a condensed float64 distance vector -> merge table [n-1, 2] (sklearn's children_)."""
    if n < 2:
        return np.zeros((0, 2), dtype=np.intp)
    return hierarchy.linkage(np.asarray(condensed, dtype=np.float64), method="ward")[:, :2].astype(np.intp)


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
    children = ward_children(condensed.cpu().numpy(), n)
    return cut_tree(children, n_clusters, n)


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


def metrics_from_labels(label, predict):
    predict = best_map(label, predict).astype(int)
    NMI = metrics.normalized_mutual_info_score(label, predict)
    ARI = metrics.adjusted_rand_score(label, predict)
    F1 = metrics.f1_score(label, predict, average="micro")
    return NMI, ARI, F1, label, predict


def cluster_metrics_from_pdist(condensed, anomaly_names):
    """The same chain from distances that are already computed (bench.py, pipeline output): condensed float64
    Euclidean distances of ALL images (tensor or array) + per-image anomaly strings.  "combined" images are
    dropped by selecting the sub-matrix (distances between the kept images do not change)."""
    from scipy.spatial.distance import squareform
    d = condensed.detach().cpu().numpy() if isinstance(condensed, torch.Tensor) else np.asarray(condensed)
    keep = [i for i, a in enumerate(anomaly_names) if a != "combined"]
    names = [anomaly_names[i] for i in keep]
    if len(keep) != len(anomaly_names):
        d = squareform(squareform(d, checks=False)[np.ix_(keep, keep)], checks=False)
    label = LabelEncoder().fit_transform(names).astype(int)
    n = len(keep)
    predict = cut_tree(ward_children(d, n), len(set(label)), n) if n > 1 else np.zeros(n, dtype=np.intp)
    return metrics_from_labels(label, predict)


def cluster_metrics(X, anomaly_names, device=None):
    """Reference :176-220 from (X, per-image anomaly strings): drop "combined", Ward with
    k = #labels, best_map, NMI / ARI / micro-F1.  Returns NMI, ARI, F1, label, predict."""
    keep = [i for i, a in enumerate(anomaly_names) if a != "combined"]
    names = [anomaly_names[i] for i in keep]
    Xk = np.asarray(X)[keep]
    label = LabelEncoder().fit_transform(names).astype(int)
    predict = agglomerative_ward(Xk, len(set(label)), device)
    return metrics_from_labels(label, predict)
