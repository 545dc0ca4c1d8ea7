"""alpha-weight operators (reference: models/patchcore/utils.py:222-277) and ViT init helper.

`Matrix_Alpha_Unsupervised(tau, k, Z, device)` / `Matrix_Alpha_Supervised(tau, k, Z, Z_train, device)`
keep their signatures and return a float64 tensor [N, P] on `device`; the O(N^2) Python loop of
per-pair `torch.cdist` calls is replaced by one fused MFMA kernel (all key patches x all query
patches, min over each key image folded into the epilogue) plus a float64 softmax kernel.
The plotting / result-folder helpers of the reference file (:18-219) are out of scope.
"""
import torch

import acnative


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Reference :216-219 (same algorithm as torch.nn.init.trunc_normal_)."""
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _on(device, t):
    return t.to(device=device, dtype=torch.float32).contiguous()


def Weight_Distance_Unsupervised(Z, i, device):
    """w_i[p] = mean_{j != i} min_q ||Z_i[p] - Z_j[q]||  (reference :222-227) -> [P] fp32."""
    Z = _on(device, Z)
    return acnative.pair_min_dist(Z[i:i + 1], Z, acnative.REDUCE_MEAN_OTHERS, self_offset=i)[0]


def Weight_Distance_Supervised(Z, Z_train, i, device):
    """w_i[p] = min_j min_q ||Z_i[p] - Ztrain_j[q]||  (reference :230-237) -> [P] fp32."""
    return acnative.pair_min_dist(_on(device, Z)[i:i + 1], _on(device, Z_train), acnative.REDUCE_MIN_ALL)[0]


def Matrix_Alpha_Unsupervised(tau, k, Z, device):
    print("{:-^80}".format("Calculating Unsupervised Alpha Matrix"))
    Z = _on(device, Z)
    w = acnative.pair_min_dist(Z, Z, acnative.REDUCE_MEAN_OTHERS)
    return acnative.alpha_softmax(w, tau, k)


def Matrix_Alpha_Supervised(tau, k, Z, Z_train, device):
    print("{:-^80}".format("Calculating Supervised Alpha Matrix"))
    w = acnative.pair_min_dist(_on(device, Z), _on(device, Z_train), acnative.REDUCE_MIN_ALL)
    return acnative.alpha_softmax(w, tau, k)
