"""ctypes binding of libac_hip.so (include/ac_hip.h) for torch device tensors.

torch is plumbing here: it owns device memory and the current HIP stream; every
function hands raw device pointers + sizes to the C ABI.  There is NO CPU
fallback and no torch-op fallback: a missing library, a CPU tensor or a non-zero
return code raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AC_LIB") or os.path.join(_HERE, "libac_hip.so")   # AC_LIB: diagnostic builds only

GEMM_AUTO, GEMM_256x128, GEMM_256x256 = 0, 1, 2
REDUCE_MEAN_OTHERS = 0
REDUCE_MIN_ALL = 1
FEAT_TOKENS = 0
FEAT_NCHW = 1

_c = ctypes
_vp, _i, _i64, _sz, _f, _d = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t, _c.c_float, _c.c_double


class BlockWeights(_c.Structure):
    """struct ac_block_weights (include/ac_hip.h)."""
    _fields_ = [(n, _vp) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                   "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


# name -> (restype, argtypes): every symbol include/ac_hip.h declares
SIGNATURES = {
    "ac_version": (_i, []),
    "ac_last_error": (_c.c_char_p, []),
    "ac_set_gemm_policy": (_i, [_i]),
    "ac_pack_hl32": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _vp]),
    "ac_vit_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ac_vit_prepare_tokens": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ac_vit_block": (_i, [_c.POINTER(BlockWeights), _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "ac_layernorm_rows": (_i, [_vp, _i64, _i, _vp, _vp, _f, _vp, _vp]),
    "ac_patch_aggregate_workspace_bytes": (_sz, [_i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _i, _i, _i, _i]),
    "ac_patch_aggregate": (_i, [_c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i),
                                _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "ac_unfold_patches": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ac_adaptive_avg_pool_rows": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "ac_pair_min_dist_workspace_bytes": (_sz, [_i64, _i64, _i, _i]),
    "ac_pair_min_dist": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i64, _vp, _vp, _sz, _vp]),
    "ac_pair_min_sq_packed": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _vp, _i, _i, _vp]),
    "ac_pair_reduce": (_i, [_vp, _i64, _i64, _i, _i, _i64, _vp, _vp]),
    "ac_alpha_softmax": (_i, [_vp, _i64, _i, _d, _d, _vp, _vp]),
    "ac_weighted_avg": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "ac_pdist": (_i, [_vp, _i64, _i, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load libac_hip.so once; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "libac_hip.so is not built (%s). Run `python __graft_entry__.py` or "
                        "`python anomaly-clustering_amd/models/acnative/build.py`. There is no CPU fallback." % LIB_PATH)
                handle = _c.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)       # AttributeError if the ABI is incomplete
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("libac_hip error %d: %s" % (rc, lib().ac_last_error().decode()))


def set_gemm_policy(policy):
    """Pin one of the two bit-identical linear-layer GEMM kernels (diagnostic; GEMM_AUTO = by size)."""
    _check(lib().ac_set_gemm_policy(int(policy)))


def _dev(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on the HIP device (got %s); the MI355X path has no CPU fallback"
                           % (name, getattr(t, "device", type(t))))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_workspaces = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer (uint8) per (device, stream): every entry point launches on the current stream,
    so calls on different streams never share scratch, and a buffer is only ever replaced by an allocation made
    on its own stream (the caching allocator then orders the reuse of the old block after the queued kernels).
    release_workspaces() gives the memory back (e.g. ~1 GB of packed operands after a large alpha call)."""
    device = torch.device(device)
    stream = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream.cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        with torch.cuda.device(device):
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces():
    _workspaces.clear()


# ------------------------------------------------------------------ operands
def pack_hl32(mat, want_sqnorm=False):
    """fp32 [rows, K] -> hl32 image (uint8 [rows, K*4]) (+ fp32 [rows] squared norms)."""
    p = _dev(mat, torch.float32, "mat")
    rows, K = mat.shape
    out = torch.empty((rows, K * 4), dtype=torch.uint8, device=mat.device)
    sq = torch.empty((rows,), dtype=torch.float32, device=mat.device) if want_sqnorm else None
    _check(lib().ac_pack_hl32(p, rows, K, K, out.data_ptr(), sq.data_ptr() if want_sqnorm else None, _stream()))
    return (out, sq) if want_sqnorm else out


# ----------------------------------------------------------------------- ViT
def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def pack_block_weights(block):
    """Device copies of one Block's parameters in the layout ac_vit_block wants."""
    keep = {}
    w = BlockWeights()

    def vec(name, t):
        keep[name] = _f32(t)
        setattr(w, name, _dev(keep[name], torch.float32, name))

    def mat(name, t):
        keep[name] = pack_hl32(_f32(t))
        setattr(w, name, keep[name].data_ptr())

    vec("ln1_w", block.norm1.weight); vec("ln1_b", block.norm1.bias)
    mat("qkv_w", block.attn.qkv.weight)
    qb = block.attn.qkv.bias
    vec("qkv_b", qb if qb is not None else torch.zeros(block.attn.qkv.out_features, device=block.attn.qkv.weight.device))
    mat("proj_w", block.attn.proj.weight); vec("proj_b", block.attn.proj.bias)
    vec("ln2_w", block.norm2.weight); vec("ln2_b", block.norm2.bias)
    mat("fc1_w", block.mlp.fc1.weight); vec("fc1_b", block.mlp.fc1.bias)
    mat("fc2_w", block.mlp.fc2.weight); vec("fc2_b", block.mlp.fc2.bias)
    return {"struct": w, "keep": keep, "eps": float(block.norm1.eps), "heads": block.attn.num_heads,
            "hidden": block.mlp.fc1.out_features}


def pack_embed_weights(model):
    pw = model.patch_embed.proj.weight
    return {"proj_w": pack_hl32(_f32(pw).reshape(pw.shape[0], -1)), "proj_b": _f32(model.patch_embed.proj.bias),
            "cls": _f32(model.cls_token).reshape(-1), "pos": _f32(model.pos_embed).reshape(-1, pw.shape[0]),
            "in_ch": pw.shape[1], "patch": pw.shape[-1]}


def _vit_ws(B, T, C, hidden, heads, patch_dim, device):
    n = lib().ac_vit_workspace_bytes(B, T, C, hidden, heads, patch_dim)
    return workspace(n, device), n


def vit_prepare_tokens(model, images):
    images = images.to(torch.float32).contiguous()
    p = _dev(images, torch.float32, "images")
    pk = model.packed()
    B, in_ch, img, _ = images.shape
    C = model.embed_dim
    T = model.pos_embed.shape[1]
    hidden = model.blocks[0].mlp.fc1.out_features
    ws, n = _vit_ws(B, T, C, hidden, model.num_heads, in_ch * pk["patch"] ** 2, images.device)
    x = torch.empty((B, T, C), dtype=torch.float32, device=images.device)
    _check(lib().ac_vit_prepare_tokens(p, B, in_ch, img, pk["patch"], C, pk["proj_w"].data_ptr(),
                                       pk["proj_b"].data_ptr(), pk["cls"].data_ptr(), pk["pos"].data_ptr(),
                                       x.data_ptr(), ws.data_ptr(), n, _stream()))
    return x


def vit_block(block, x):
    x = x.contiguous()
    p = _dev(x, torch.float32, "x")
    pk = block.packed()
    B, T, C = x.shape
    ws, n = _vit_ws(B, T, C, pk["hidden"], pk["heads"], 32, x.device)
    y = torch.empty_like(x)
    _check(lib().ac_vit_block(_c.byref(pk["struct"]), p, y.data_ptr(), B, T, C, pk["heads"], pk["hidden"],
                              pk["eps"], ws.data_ptr(), n, _stream()))
    return y


def layernorm_rows(x, norm):
    x = x.contiguous()
    p = _dev(x, torch.float32, "x")
    rows, C = x.shape
    w, b = _f32(norm.weight), _f32(norm.bias)
    y = torch.empty_like(x)
    _check(lib().ac_layernorm_rows(p, rows, C, w.data_ptr(), b.data_ptr(), float(norm.eps), y.data_ptr(), _stream()))
    return y


# --------------------------------------------------------- patch aggregation
def patch_aggregate(features, patchsize, pre_dim, tgt_dim, eps=1e-5):
    """features: list of fp32 device tensors, each [B, T, C] (token map with CLS) or [B, C, H, W].
    Returns Z fp32 [B * H0 * W0, tgt_dim] and the per-layer patch grids."""
    n = len(features)
    feats = [f.to(torch.float32).contiguous() for f in features]
    ptrs = (_vp * n)(*[_dev(f, torch.float32, "feature") for f in feats])
    kinds, Cs, Hs, Ws = (_i * n)(), (_i * n)(), (_i * n)(), (_i * n)()
    B = feats[0].shape[0]
    for j, f in enumerate(feats):
        if f.dim() == 3:
            side = int(round((f.shape[1] - 1) ** 0.5))
            if side * side + 1 != f.shape[1]:
                raise ValueError("token map with %d tokens is not CLS + square grid" % f.shape[1])
            kinds[j], Cs[j], Hs[j], Ws[j] = FEAT_TOKENS, f.shape[2], side, side
        elif f.dim() == 4:
            kinds[j], Cs[j], Hs[j], Ws[j] = FEAT_NCHW, f.shape[1], f.shape[2], f.shape[3]
        else:
            raise ValueError("feature map must be [B,T,C] or [B,C,H,W]")
        if f.shape[0] != B:
            raise ValueError("feature maps disagree on batch size")
    nbytes = lib().ac_patch_aggregate_workspace_bytes(n, Cs, Hs, Ws, B, patchsize, pre_dim, tgt_dim)
    ws = workspace(nbytes, feats[0].device)
    Z = torch.empty((B * Hs[0] * Ws[0], tgt_dim), dtype=torch.float32, device=feats[0].device)
    _check(lib().ac_patch_aggregate(ptrs, kinds, Cs, Hs, Ws, n, B, patchsize, pre_dim, tgt_dim, float(eps),
                                    Z.data_ptr(), ws.data_ptr(), nbytes, _stream()))
    return Z, [[Hs[j], Ws[j]] for j in range(n)]


def unfold_patches(x, patchsize, stride=1):
    """[B, C, H, W] -> [B, H*W, C, k, k], zero padded (PatchMaker.patchify)."""
    if stride != 1 or patchsize % 2 == 0:
        raise NotImplementedError("only odd patch sizes at stride 1 are on the anomaly-clustering path")
    x = x.to(torch.float32).contiguous()
    p = _dev(x, torch.float32, "features")
    B, C, H, W = x.shape
    out = torch.empty((B, H * W, C, patchsize, patchsize), dtype=torch.float32, device=x.device)
    _check(lib().ac_unfold_patches(p, B, C, H, W, patchsize, out.data_ptr(), _stream()))
    return out


def adaptive_pool_rows(x, out_dim):
    """fp32 [rows, L] -> [rows, out_dim] with torch's adaptive_avg_pool1d windows."""
    p = _dev(x, torch.float32, "x")
    rows, L = x.shape
    y = torch.empty((rows, out_dim), dtype=torch.float32, device=x.device)
    _check(lib().ac_adaptive_avg_pool_rows(p, rows, L, out_dim, y.data_ptr(), _stream()))
    return y


# --------------------------------------------------------------------- alpha
def pair_min_dist(Zq, Zk, reduce, self_offset=0):
    """w[i][p] = reduce_j min_q ||Zq[i][p] - Zk[j][q]||  ->  fp32 [nq, P]."""
    pq, pk = _dev(Zq, torch.float32, "Zq"), _dev(Zk, torch.float32, "Zk")
    nq, P, D = Zq.shape
    nk = Zk.shape[0]
    if Zk.shape[1:] != Zq.shape[1:]:
        raise ValueError("Zq / Zk patch shapes differ: %s vs %s" % (tuple(Zq.shape), tuple(Zk.shape)))
    same = pq == pk and nq == nk
    nbytes = lib().ac_pair_min_dist_workspace_bytes(nq, nk, P, D)
    ws = workspace(nbytes, Zq.device)
    w = torch.empty((nq, P), dtype=torch.float32, device=Zq.device)
    _check(lib().ac_pair_min_dist(pq, nq, pq if same else pk, nk, P, D, reduce, self_offset, w.data_ptr(),
                                  ws.data_ptr(), nbytes, _stream()))
    return w


kernel_events = None   # set to a list to collect (label, start, stop) torch events around MFMA launches


def pair_min_dist_staged(Zq, Zk, reduce, self_offset=0, minbuf_budget=1 << 30):
    """Same result as pair_min_dist, composed from the staged entry points (pack once, one MFMA
    launch per query chunk, reduce).  When `kernel_events` is a list, each MFMA launch is bracketed
    by events recorded on the launch stream (bench.py's roofline leg reads them)."""
    pq, pk = _dev(Zq, torch.float32, "Zq"), _dev(Zk, torch.float32, "Zk")
    nq, P, D = Zq.shape
    nk = Zk.shape[0]
    if Zk.shape[1:] != Zq.shape[1:]:
        raise ValueError("Zq / Zk patch shapes differ: %s vs %s" % (tuple(Zq.shape), tuple(Zk.shape)))
    hk, sk = pack_hl32(Zk.reshape(nk * P, D), want_sqnorm=True)
    if pq == pk and nq == nk:
        hq, sq = hk, sk
    elif pk <= pq < pk + Zk.numel() * 4 and (pq - pk) % (P * D * 4) == 0 and pq + Zq.numel() * 4 <= pk + Zk.numel() * 4:
        first = (pq - pk) // (P * D * 4)                      # the query shard is a slice of the key set
        hq, sq = hk[first * P:(first + nq) * P], sk[first * P:(first + nq) * P]
    else:
        hq, sq = pack_hl32(Zq.reshape(nq * P, D), want_sqnorm=True)
    chunk = max(1, min(nq, minbuf_budget // (nk * P * 4)))
    w = torch.empty((nq, P), dtype=torch.float32, device=Zq.device)
    minbuf = torch.empty((nk, chunk * P), dtype=torch.int32, device=Zq.device)
    st = _stream()
    for q0 in range(0, nq, chunk):
        qn = min(chunk, nq - q0)
        mb = minbuf.view(-1)[: nk * qn * P]
        mb.fill_(0x7f800000)
        if kernel_events is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        _check(lib().ac_pair_min_sq_packed(hk.data_ptr(), sk.data_ptr(), nk, hq[q0 * P:].data_ptr(),
                                           sq[q0 * P:].data_ptr(), qn, P, D, mb.data_ptr(), 0, 1, st))
        if kernel_events is not None:
            ev1.record()
            kernel_events.append(("pair_min_sq", ev0, ev1, nk, qn, P, D))
        _check(lib().ac_pair_reduce(mb.data_ptr(), nk, qn, P, reduce, self_offset + q0, w[q0:].data_ptr(), st))
    return w


def pair_min_sq_part(Z, part=0, nparts=1):
    """One slice of the SYMMETRIC all-pairs problem on Z [N, P, D] (keys == queries): returns
    minbuf fp32 [N, N*P] with minbuf[j][i*P + p] = min_q ||Z[i][p] - Z[j][q]||^2 for the (i, j) tile
    pairs of this slice and +inf elsewhere.  The element-wise minimum over all `nparts` slices (e.g.
    a MIN all-reduce across ranks) is the complete matrix; `pair_reduce` turns it into w."""
    _dev(Z, torch.float32, "Z")
    N, P, D = Z.shape
    hk, sk = pack_hl32(Z.reshape(N * P, D), want_sqnorm=True)
    minbuf = torch.full((N, N * P), float("inf"), dtype=torch.float32, device=Z.device)
    if kernel_events is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _check(lib().ac_pair_min_sq_packed(hk.data_ptr(), sk.data_ptr(), N, hk.data_ptr(), sk.data_ptr(), N, P, D,
                                       minbuf.data_ptr(), part, nparts, _stream()))
    if kernel_events is not None:
        ev1.record()
        kernel_events.append(("pair_min_sq", ev0, ev1, N, N / float(nparts), P, D))
    return minbuf


def pair_reduce(minbuf, P, reduce, self_first=0):
    """minbuf fp32 [nk, nq*P] (squared mins) -> w fp32 [nq, P]: mean over key images != self_first + i
    of the square roots, or the square root of the min over all key images."""
    p = _dev(minbuf, torch.float32, "minbuf")
    nk, cols = minbuf.shape
    nq = cols // P
    w = torch.empty((nq, P), dtype=torch.float32, device=minbuf.device)
    if nq == 0:                      # a rank with an empty image shard
        return w
    _check(lib().ac_pair_reduce(p, nk, nq, P, reduce, self_first, w.data_ptr(), _stream()))
    return w


def alpha_softmax(w, tau, k=1.0):
    p = _dev(w, torch.float32, "w")
    N, P = w.shape
    a = torch.empty((N, P), dtype=torch.float64, device=w.device)
    if N == 0:
        return a
    _check(lib().ac_alpha_softmax(p, N, P, float(tau), float(k), a.data_ptr(), _stream()))
    return a


def weighted_avg(alpha, Z):
    """alpha fp32 [N, P] (or [N,1,P]), Z fp32 [N, P, D] -> X fp32 [N, D]."""
    N, P, D = Z.shape
    if N == 0:                           # a rank with an empty image shard
        return torch.empty((0, D), dtype=torch.float32, device=Z.device)
    alpha = alpha.reshape(alpha.shape[0], -1).contiguous()
    pa, pz = _dev(alpha, torch.float32, "alpha"), _dev(Z, torch.float32, "Z")
    if alpha.shape != (N, P):
        raise ValueError("alpha %s does not match Z %s" % (tuple(alpha.shape), tuple(Z.shape)))
    X = torch.empty((N, D), dtype=torch.float32, device=Z.device)
    for s in range(0, N, 32768):
        e = min(N, s + 32768)
        _check(lib().ac_weighted_avg(alpha[s:e].data_ptr(), Z[s:e].data_ptr(), e - s, P, D, X[s:e].data_ptr(),
                                     _stream()))
    return X


def pdist(X):
    """Condensed float64 Euclidean distances of fp32 X [N, D] (scipy pdist order)."""
    p = _dev(X, torch.float32, "X")
    N, D = X.shape
    out = torch.empty((N * (N - 1) // 2,), dtype=torch.float64, device=X.device)
    _check(lib().ac_pdist(p, N, D, out.data_ptr(), _stream()))
    return out
