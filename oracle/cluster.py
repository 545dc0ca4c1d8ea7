"""Oracle: weighted-average embeddings -> Ward clustering -> NMI / ARI / F1.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Restates examples/test.py:109-131 (best_map) and :176-220 (calculate_metrics
after the pickle load; the matplotlib heat-map block :158-174 is out of
scope).  Third-party arithmetic named by the reference's requirements.txt:
scikit_learn==1.1.3 (AgglomerativeClustering -> ward_tree ->
scipy.cluster.hierarchy.ward; metrics), scipy==1.10.0, munkres==1.1.4.
munkres is absent here; scipy.optimize.linear_sum_assignment solves the same
assignment problem and reproduces all 26 committed known-answer rows
(tests/golden/kat_*.npz) to <= 1.2e-16 with sklearn 1.7.2 / scipy 1.15.3.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import pdist as _pdist
from sklearn import cluster, metrics
from sklearn.preprocessing import LabelEncoder


def best_map(L1, L2):
    """test.py:109-131.  L1 = ground-truth labels, L2 = cluster ids; relabels
    L2 with the L1 label that maximises the matched count (Hungarian on -G.T)."""
    Label1 = np.unique(L1)
    Label2 = np.unique(L2)
    nClass1, nClass2 = len(Label1), len(Label2)
    nClass = max(nClass1, nClass2)
    G = np.zeros((nClass, nClass))
    for i in range(nClass1):
        for j in range(nClass2):
            G[i, j] = np.sum((L2 == Label2[j]).astype(float) * (L1 == Label1[i]).astype(float))
    rows, cols = linear_sum_assignment(-G.T)   # Munkres().compute(-G.T) -> [(row, col)]
    c = cols[np.argsort(rows)]
    newL2 = np.zeros(L2.shape)
    for i in range(nClass2):
        newL2[L2 == Label2[i]] = Label1[c[i]]
    return newL2


def drop_combined(X, anomaly_names):
    """test.py:176-188: remove multi-label ("combined") images; X -> float64
    (the reference builds X with np.append onto a float64 zeros row)."""
    keep = [i for i, a in enumerate(anomaly_names) if a != "combined"]
    return np.asarray(X)[keep].astype(np.float64), [anomaly_names[i] for i in keep], keep


def calculate_metrics(X, anomaly_names):
    """test.py:176-220 given X [N, D] and the per-image anomaly-type strings.
    Returns NMI, ARI, F1(micro), label, predict."""
    X64, names, _ = drop_combined(X, anomaly_names)
    label = LabelEncoder().fit_transform(names).astype(int)
    model = cluster.AgglomerativeClustering(n_clusters=len(set(label)))
    predict = best_map(label, model.fit_predict(X64)).astype(int)
    NMI = metrics.normalized_mutual_info_score(label, predict)
    ARI = metrics.adjusted_rand_score(label, predict)
    F1 = metrics.f1_score(label, predict, average="micro")
    return NMI, ARI, F1, label, predict


def calculate_metrics_from_pdist(condensed, names):
    """test.py:190-220 from the point where sklearn's AgglomerativeClustering hands over to scipy: ward_tree without
    connectivity is scipy.cluster.hierarchy.ward on the float64 rows (= linkage(pdist(X), "ward")), children =
    the first two columns, labels = sklearn's _hc_cut.  `condensed`: scipy-order float64 distances of the kept images
    (tests/golden/kat_all_pdist.npz), `names`: their anomaly-type strings.  Returns NMI, ARI, F1, label, predict, Z."""
    from scipy.cluster import hierarchy
    from sklearn.cluster._agglomerative import _hc_cut
    label = LabelEncoder().fit_transform(list(names)).astype(int)
    n = len(label)
    Z = hierarchy.linkage(np.asarray(condensed, dtype=np.float64), method="ward")
    children = Z[:, :2].astype(np.intp)
    predict = best_map(label, _hc_cut(len(set(label)), children, n)).astype(int)
    NMI = metrics.normalized_mutual_info_score(label, predict)
    ARI = metrics.adjusted_rand_score(label, predict)
    F1 = metrics.f1_score(label, predict, average="micro")
    return NMI, ARI, F1, label, predict, Z


def pdist(X):
    """scipy pdist as called inside scipy.cluster.hierarchy.ward (via sklearn
    ward_tree, test.py:193-195): condensed Euclidean distances in float64."""
    return _pdist(np.asarray(X, dtype=np.float64), metric="euclidean")
