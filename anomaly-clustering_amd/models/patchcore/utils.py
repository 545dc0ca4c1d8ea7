"""alpha-weight operators (reference: models/patchcore/utils.py:222-277) and ViT init helper.

`Matrix_Alpha_Unsupervised(tau, k, Z, device)` / `Matrix_Alpha_Supervised(tau, k, Z, Z_train, device)`
keep their signatures and return a float64 tensor [N, P] on `device`; the O(N^2) Python loop of
per-pair `torch.cdist` calls is replaced by the fused MFMA kernel (all key patches x all query
patches, min over each key image folded into the epilogue) driven by the block schedule of
patchcore.pipeline - one symmetric launch for an MVTec-sized category, sub-blocks inside a fixed memory
budget for a large one - plus a float64 softmax kernel.
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


def _single_rank_ops():
    from patchcore.pipeline import HipOps
    return HipOps(core=None)


def Matrix_Alpha_Unsupervised(tau, k, Z, device):
    print("{:-^80}".format("Calculating Unsupervised Alpha Matrix"))
    from patchcore.pipeline import alpha_weights_unsupervised
    Z = _on(device, Z)
    ops = _single_rank_ops()
    w = alpha_weights_unsupervised(ops, ops.pack(Z), [(0, Z.shape[0])], 0, 1, None, Z.device)
    return acnative.alpha_softmax(w, tau, k)


def Matrix_Alpha_Supervised(tau, k, Z, Z_train, device):
    print("{:-^80}".format("Calculating Supervised Alpha Matrix"))
    from patchcore.pipeline import alpha_weights_supervised
    Z, Z_train = _on(device, Z), _on(device, Z_train)
    ops = _single_rank_ops()
    w = alpha_weights_supervised(ops, ops.pack(Z), ops.pack(Z_train), [(0, Z_train.shape[0])], 0, 1, None, Z.device)
    return acnative.alpha_softmax(w, tau, k)
