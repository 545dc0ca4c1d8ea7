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
FEAT_NHWC = 2

_c = ctypes
_vp, _i, _i64, _sz, _f, _d = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t, _c.c_float, _c.c_double


class BlockWeights(_c.Structure):
    """struct ac_block_weights (include/ac_hip.h)."""
    _fields_ = [(n, _vp) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                   "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


ABI_VERSION = 600      # include/ac_hip.h AC_ABI_VERSION

# name -> (restype, argtypes): every symbol include/ac_hip.h declares
SIGNATURES = {
    "ac_version": (_i, []),
    "ac_last_error": (_c.c_char_p, []),
    "ac_set_gemm_policy": (_i, [_i]),
    "ac_pack_hl32": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _vp]),
    "ac_preprocess_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ac_preprocess_images": (_i, [_vp, _i, _i, _i, _i, _i, _c.POINTER(_f), _c.POINTER(_f), _vp, _vp, _sz, _vp]),
    "ac_vit_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ac_vit_prepare_tokens": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ac_vit_block": (_i, [_c.POINTER(BlockWeights), _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "ac_layernorm_rows": (_i, [_vp, _i64, _i, _vp, _vp, _f, _vp, _vp]),
    "ac_im2col_hl32": (_i, [_vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "ac_conv_gemm_bn": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "ac_maxpool_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ac_patch_aggregate_workspace_bytes": (_sz, [_i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _i, _i, _i, _i]),
    "ac_patch_aggregate": (_i, [_c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i),
                                _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "ac_patch_aggregate_hl32_supported": (_i, [_c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _i, _i, _i, _i]),
    "ac_patch_aggregate_hl32": (_i, [_c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i),
                                     _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "ac_unfold_patches": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ac_adaptive_avg_pool_rows": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "ac_pair_min_dist_workspace_bytes": (_sz, [_i64, _i64, _i, _i]),
    "ac_pair_min_dist": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i64, _vp, _vp, _sz, _vp]),
    "ac_pair_min_sq_packed": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _vp, _vp, _i, _i, _vp]),
    "ac_fill_u32": (_i, [_vp, _i64, _c.c_uint32, _vp]),
    "ac_fill_u64": (_i, [_vp, _i64, _c.c_uint64, _vp]),
    "ac_max_f32": (_i, [_vp, _i64, _vp, _vp]),
    "ac_pair_fixed_shift": (_i, [_vp, _i64, _vp, _vp]),
    "ac_pair_accumulate": (_i, [_vp, _i64, _i64, _i64, _i, _i, _i64, _vp, _vp, _vp]),
    "ac_pair_finalize": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _vp]),
    "ac_h16_from_hl32": (_i, [_vp, _i64, _i, _vp, _vp]),
    "ac_nn_slack": (_i, [_vp, _vp, _vp, _vp]),
    "ac_nn_candidates": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "ac_nn_refine": (_i, [_vp, _i64, _vp, _i64, _i, _vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ac_nn_overflow_compact": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "ac_gather_rows": (_i, [_vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "ac_nn_scatter_exhaustive": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "ac_nn_finalize": (_i, [_vp, _i64, _vp, _vp]),
    "ac_alpha_softmax": (_i, [_vp, _i64, _i, _d, _d, _vp, _vp]),
    "ac_alpha_softmax_f32": (_i, [_vp, _i64, _i, _d, _d, _vp, _vp, _vp]),
    "ac_weighted_avg": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "ac_pdist": (_i, [_vp, _i64, _i, _vp, _vp]),
    "ac_ward_workspace_bytes": (_sz, [_i64]),
    "ac_ward_linkage": (_i, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ac_cluster_scores": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
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
                if handle.ac_version() != ABI_VERSION:
                    raise RuntimeError("%s has ABI revision %d, this binding is written for %d (include/ac_hip.h "
                                       "AC_ABI_VERSION): rebuild the library" % (LIB_PATH, handle.ac_version(), ABI_VERSION))
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
def pack_hl32(mat, want_sqnorm=False, out=None, sq_out=None):
    """fp32 [rows, K] -> hl32 image (uint8 [rows, K*4]) (+ fp32 [rows] squared norms).  out / sq_out: contiguous
    device tensors of those shapes to write into (a block of rows of a larger image)."""
    p = _dev(mat, torch.float32, "mat")
    rows, K = mat.shape
    if out is None:
        out = torch.empty((rows, K * 4), dtype=torch.uint8, device=mat.device)
    else:
        _dev(out, torch.uint8, "out")
        if tuple(out.shape) != (rows, K * 4):
            raise ValueError("pack_hl32: out must be uint8 [%d, %d], got %s" % (rows, K * 4, tuple(out.shape)))
    sq = None
    if want_sqnorm:
        sq = torch.empty((rows,), dtype=torch.float32, device=mat.device) if sq_out is None else sq_out
        if tuple(sq.shape) != (rows,):
            raise ValueError("pack_hl32: sq_out must be fp32 [%d], got %s" % (rows, tuple(sq.shape)))
        _dev(sq, torch.float32, "sq_out")
    _check(lib().ac_pack_hl32(p, rows, K, K, out.data_ptr(), sq.data_ptr() if want_sqnorm else None, _stream()))
    return (out, sq) if want_sqnorm else out


# ------------------------------------------------------------ input pipeline
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_images(images_u8, resize=256, crop=224, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 device tensor [B, H, W, 3] (decoded RGB) -> fp32 [B, 3, crop, crop]: Resize / CenterCrop /
    ToTensor / Normalize of mvtec.py:74-81 in one batched call, bit-identical to the Pillow + torch host chain."""
    p = _dev(images_u8, torch.uint8, "images")
    if images_u8.dim() != 4 or images_u8.shape[3] != 3:
        raise ValueError("images must be uint8 [B, H, W, 3], got %s" % (tuple(images_u8.shape),))
    B, H, W, _ = images_u8.shape
    out = torch.empty((B, 3, crop, crop), dtype=torch.float32, device=images_u8.device)
    if B == 0:
        return out
    n = lib().ac_preprocess_workspace_bytes(B, H, W, resize, crop)
    ws = workspace(n, images_u8.device)
    m, s = (_f * 3)(*[float(v) for v in mean]), (_f * 3)(*[float(v) for v in std])
    _check(lib().ac_preprocess_images(p, B, H, W, resize, crop, m, s, out.data_ptr(), ws.data_ptr(), n, _stream()))
    return out


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


# ------------------------------------------------------ convolutional layers
def im2col_hl32(x_nhwc, kh, kw, stride, pad, Kpad):
    """x: fp32 device tensor indexed [B, H, W, C] with ANY strides (NHWC activations, or the permuted view of the NCHW
    image) -> hl32 uint8 [B*Ho*Wo, Kpad*4], column k = (ky*kw + kx)*C + c (include/ac_hip.h)."""
    if not isinstance(x_nhwc, torch.Tensor) or not x_nhwc.is_cuda or x_nhwc.dtype != torch.float32 or x_nhwc.dim() != 4:
        raise RuntimeError("im2col_hl32 needs an fp32 [B, H, W, C] tensor on the HIP device; there is no CPU fallback")
    B, H, W, C = x_nhwc.shape
    sb, sy, sx, sc = x_nhwc.stride()
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = torch.empty((B * Ho * Wo, Kpad * 4), dtype=torch.uint8, device=x_nhwc.device)
    _check(lib().ac_im2col_hl32(x_nhwc.data_ptr(), B, C, H, W, sb, sc, sy, sx, kh, kw, stride, pad, Kpad, out.data_ptr(), _stream()))
    return out


def conv_gemm_bn(A_hl, W_hl, M, N, K, alpha, beta, res, relu):
    """fp32 [M, N] = relu?((A . W^T) * alpha + beta (+ res)): convolution as a GEMM + eval BatchNorm + residual + ReLU."""
    out = torch.empty((M, N), dtype=torch.float32, device=A_hl.device)
    if A_hl.numel() != M * K * 4 or W_hl.numel() != N * K * 4:
        raise ValueError("operand sizes do not match M=%d N=%d K=%d" % (M, N, K))
    _check(lib().ac_conv_gemm_bn(_dev(A_hl, torch.uint8, "A"), _dev(W_hl, torch.uint8, "W"), M, N, K,
                                 _dev(alpha, torch.float32, "alpha"), _dev(beta, torch.float32, "beta"),
                                 _dev(res, torch.float32, "res") if res is not None else None, int(bool(relu)),
                                 out.data_ptr(), _stream()))
    return out


def maxpool_nhwc(x, k, stride, pad):
    """nn.MaxPool2d(k, stride, pad) on a contiguous fp32 [B, H, W, C] device tensor."""
    p = _dev(x, torch.float32, "x")
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    _check(lib().ac_maxpool_nhwc(p, B, H, W, C, k, stride, pad, out.data_ptr(), _stream()))
    return out


# --------------------------------------------------------- patch aggregation
def _aggregate_layout(features):
    """The feature maps as the library takes them: (tensors kept alive, ptrs, kinds, Cs, Hs, Ws, n, B)."""
    n = len(features)
    feats, nhwc = [], []
    for f in features:
        # a [B, C, H, W] tensor in channels-last memory (what the HIP CNN layers write) is read in place
        cl = f.dim() == 4 and f.dtype == torch.float32 and not f.is_contiguous() and f.permute(0, 2, 3, 1).is_contiguous()
        nhwc.append(cl)
        feats.append(f.permute(0, 2, 3, 1) if cl else f.to(torch.float32).contiguous())
    ptrs = (_vp * n)(*[_dev(f, torch.float32, "feature") for f in feats])
    kinds, Cs, Hs, Ws = (_i * n)(), (_i * n)(), (_i * n)(), (_i * n)()
    B = feats[0].shape[0]
    for j, f in enumerate(feats):
        if nhwc[j]:
            kinds[j], Cs[j], Hs[j], Ws[j] = FEAT_NHWC, f.shape[3], f.shape[1], f.shape[2]
        elif f.dim() == 3:
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
    return feats, ptrs, kinds, Cs, Hs, Ws, n, B


def patch_aggregate(features, patchsize, pre_dim, tgt_dim, eps=1e-5, out=None):
    """features: list of fp32 device tensors, each [B, T, C] (token map with CLS) or [B, C, H, W].
    Returns Z fp32 [B * H0 * W0, tgt_dim] (written into `out` when given) and the per-layer patch grids."""
    feats, ptrs, kinds, Cs, Hs, Ws, n, B = _aggregate_layout(features)
    nbytes = lib().ac_patch_aggregate_workspace_bytes(n, Cs, Hs, Ws, B, patchsize, pre_dim, tgt_dim)
    ws = workspace(nbytes, feats[0].device)
    if out is None:
        Z = torch.empty((B * Hs[0] * Ws[0], tgt_dim), dtype=torch.float32, device=feats[0].device)
    else:
        _dev(out, torch.float32, "out")
        if out.numel() != B * Hs[0] * Ws[0] * tgt_dim or out.device != feats[0].device:
            raise ValueError("out has %d elements, the batch needs %d" % (out.numel(), B * Hs[0] * Ws[0] * tgt_dim))
        Z = out.view(B * Hs[0] * Ws[0], tgt_dim)
    _check(lib().ac_patch_aggregate(ptrs, kinds, Cs, Hs, Ws, n, B, patchsize, pre_dim, tgt_dim, float(eps),
                                    Z.data_ptr(), ws.data_ptr(), nbytes, _stream()))
    return Z, [[Hs[j], Ws[j]] for j in range(n)]


def patch_aggregate_hl32_supported(features, patchsize, pre_dim, tgt_dim):
    """Can ac_patch_aggregate_hl32 take these maps (the token-map shape of the register path: DINO ViT-B/8)?"""
    if any(f.dim() != 3 or f.dtype != torch.float32 for f in features):
        return False
    _, _, kinds, Cs, Hs, Ws, n, _ = _aggregate_layout(features)
    return bool(lib().ac_patch_aggregate_hl32_supported(kinds, Cs, Hs, Ws, n, patchsize, pre_dim, tgt_dim))


def patch_aggregate_hl32(features, patchsize, pre_dim, tgt_dim, out_hl, eps=1e-5):
    """The aggregation with its rows written straight into the hl32 rows `out_hl` (uint8 [B * H0 * W0, tgt_dim * 4], e.g.
    Packed.hl of a block of B images): byte for byte pack_images(patch_aggregate(...)).hl, without the fp32 Z in between.
    The squared norms are NOT produced (a received shard's norms travel with its taps)."""
    feats, ptrs, kinds, Cs, Hs, Ws, n, B = _aggregate_layout(features)
    rows = B * Hs[0] * Ws[0]
    _dev(out_hl, torch.uint8, "out_hl")
    if out_hl.numel() != rows * tgt_dim * 4 or out_hl.device != feats[0].device:
        raise ValueError("out_hl has %d bytes, the batch needs %d" % (out_hl.numel(), rows * tgt_dim * 4))
    nbytes = lib().ac_patch_aggregate_workspace_bytes(n, Cs, Hs, Ws, B, patchsize, pre_dim, tgt_dim)
    ws = workspace(nbytes, feats[0].device)
    _check(lib().ac_patch_aggregate_hl32(ptrs, kinds, Cs, Hs, Ws, n, B, patchsize, pre_dim, tgt_dim, float(eps),
                                         out_hl.data_ptr(), ws.data_ptr(), nbytes, _stream()))
    return out_hl


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


kernel_events = None   # set to a list to collect (label, start, stop, ...) torch events around MFMA launches
NO_SELF = -(1 << 62)   # self_first value that skips nothing in pair_accumulate
INF_BITS = 0x7f800000
F64_INF_BITS = 0x7ff0000000000000


class Packed:
    """hl32 image of a block of images: hl uint8 [n*P, D*4], sq fp32 [n*P] (squared row norms)."""

    def __init__(self, hl, sq, n, P, D):
        self.hl, self.sq, self.n, self.P, self.D = hl, sq, n, P, D

    def slice(self, lo, hi):
        """Images [lo, hi) as views (rows are image-major, so a block of images is a block of rows)."""
        P = self.P
        return Packed(self.hl[lo * P: hi * P], self.sq[lo * P: hi * P], hi - lo, P, self.D)

    def tensors(self):
        return [self.hl, self.sq]

    @staticmethod
    def empty(n, P, D, device):
        return Packed(torch.empty((n * P, D * 4), dtype=torch.uint8, device=device),
                      torch.empty((n * P,), dtype=torch.float32, device=device), n, P, D)


def pack_images(Z, out=None):
    """Z fp32 [n, P, D] -> Packed (one pass: split-fp16 image + squared norms).  out: a Packed of n images (e.g. a
    slice of a larger one) that receives the rows instead of a new allocation."""
    _dev(Z, torch.float32, "Z")
    n, P, D = Z.shape
    if out is not None and (out.n, out.P, out.D) != (n, P, D):
        raise ValueError("pack_images: out holds %s, Z is %s" % ((out.n, out.P, out.D), (n, P, D)))
    if n == 0:
        return out if out is not None else Packed.empty(0, P, D, Z.device)
    if out is not None:
        pack_hl32(Z.reshape(n * P, D), want_sqnorm=True, out=out.hl, sq_out=out.sq)
        return out
    hl, sq = pack_hl32(Z.reshape(n * P, D), want_sqnorm=True)
    return Packed(hl, sq, n, P, D)


def max_sq(packed_list, device, out=None):
    """Running max of the squared norms over several packed blocks: fp32 [1] on the device (`out`: a 1-element view
    of a vector the caller reduces across ranks in one collective)."""
    out = filled((1,), torch.float32, device, 0) if out is None else fill_(out, 0)
    for pk in packed_list:
        if pk.n:
            _check(lib().ac_max_f32(pk.sq.data_ptr(), pk.sq.numel(), out.data_ptr(), _stream()))
    return out


def fixed_shift(max_sq_t, n_keys):
    """int32 [1] on the device: the fixed-point scale of the unsupervised sum over n_keys key images."""
    shift = torch.empty(1, dtype=torch.int32, device=max_sq_t.device)
    _check(lib().ac_pair_fixed_shift(_dev(max_sq_t, torch.float32, "max_sq"), int(n_keys), shift.data_ptr(), _stream()))
    return shift


def fill_(t, bits=0):
    """In place: every element of the contiguous device tensor t (4- or 8-byte elements) = the bit pattern `bits`,
    written by the LIBRARY on the current stream (ac_fill_u32 / ac_fill_u64)."""
    if t.numel():
        _dev(t, None, "t")
        if t.element_size() == 4:
            _check(lib().ac_fill_u32(t.data_ptr(), t.numel(), bits, _stream()))
        elif t.element_size() == 8:
            _check(lib().ac_fill_u64(t.data_ptr(), t.numel(), bits, _stream()))
        else:
            raise TypeError("fill_: %s has neither 4- nor 8-byte elements" % t.dtype)
    return t


def filled(shape, dtype, device, bits=0):
    """A new device tensor preset to a bit pattern by the library: +inf float bits for matrices of minima, 0 for
    counters and int64 sums.  The engine's loops contain no torch kernels - only allocations."""
    return fill_(torch.empty(shape, dtype=dtype, device=device), bits)


def new_acc(ncols, reduce, device):
    """Accumulator of pair_accumulate: int64 zeros (mean) or uint32-as-int32 +inf bits (min)."""
    if reduce == REDUCE_MEAN_OTHERS:
        return filled((ncols,), torch.int64, device, 0)
    return filled((ncols,), torch.int32, device, INF_BITS)


def _event_pair():
    if kernel_events is None:
        return None, None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    return ev0, ev1


def pair_block(keys, queries, two_way=False, part=0, nparts=1):
    """One MFMA launch on the rectangle keys x queries (Packed blocks).
    Returns minbuf fp32-bits [nk, nq*P] (and minbuf_t [nq, nk*P] when two_way).  keys is queries (same
    buffers): the symmetric case - upper-triangle tiles only, optionally the part-th of nparts slices."""
    nk, nq, P, D = keys.n, queries.n, keys.P, keys.D
    if queries.P != P or queries.D != D:
        raise ValueError("key / query blocks disagree on the patch shape")
    dev = keys.hl.device
    same = keys.hl.data_ptr() == queries.hl.data_ptr() and nk == nq
    if same and two_way:
        raise ValueError("keys == queries is the symmetric case, not a two-way block")
    minbuf = filled((nk, nq * P), torch.int32, dev, INF_BITS)
    minbuf_t = filled((nq, nk * P), torch.int32, dev, INF_BITS) if two_way else None
    ev0, ev1 = _event_pair()
    _check(lib().ac_pair_min_sq_packed(keys.hl.data_ptr(), keys.sq.data_ptr(), nk, queries.hl.data_ptr(),
                                       queries.sq.data_ptr(), nq, P, D, minbuf.data_ptr(),
                                       minbuf_t.data_ptr() if two_way else None, part, nparts, _stream()))
    if ev0 is not None:
        ev1.record()
        # ordered image pairs this launch resolves: both directions of a symmetric / two-way block
        pairs = (nk * (nk - 1.0) / nparts) if same else (2.0 * nk * nq if two_way else 1.0 * nk * nq)
        kernel_events.append(("pair_min_sq", ev0, ev1, pairs, P, D))
    return (minbuf, minbuf_t) if two_way else minbuf


def pair_accumulate(minbuf, P, reduce, self_first, shift, acc):
    """Fold minbuf [nk, ncols] (rows may be strided) into acc (a slice of new_acc()'s tensor)."""
    nk, ncols = minbuf.shape
    if minbuf.stride(1) != 1:
        raise ValueError("minbuf columns must be contiguous")
    if acc.numel() != ncols or not acc.is_contiguous():
        raise ValueError("acc must be a contiguous slice of %d columns" % ncols)
    if nk == 0 or ncols == 0:
        return
    _check(lib().ac_pair_accumulate(minbuf.data_ptr(), nk, minbuf.stride(0), ncols, P, reduce, int(self_first),
                                    shift.data_ptr() if shift is not None else None, acc.data_ptr(), _stream()))


def pair_finalize(acc, P, reduce, count, shift):
    """acc [n*P] -> w fp32 [n, P]."""
    ncols = acc.numel()
    w = torch.empty((ncols // P, P), dtype=torch.float32, device=acc.device)
    if ncols:
        _check(lib().ac_pair_finalize(acc.data_ptr(), ncols, reduce, int(count),
                                      shift.data_ptr() if shift is not None else None, w.data_ptr(), _stream()))
    return w


def pair_min_dist_staged(Zq, Zk, reduce, self_offset=0, minbuf_budget=1 << 30):
    """Same result as pair_min_dist, composed from the staged entry points (pack once, one MFMA launch per
    query chunk, accumulate, finalize)."""
    pq, pk = _dev(Zq, torch.float32, "Zq"), _dev(Zk, torch.float32, "Zk")
    nq, P, D = Zq.shape
    nk = Zk.shape[0]
    if Zk.shape[1:] != Zq.shape[1:]:
        raise ValueError("Zq / Zk patch shapes differ: %s vs %s" % (tuple(Zq.shape), tuple(Zk.shape)))
    keys = pack_images(Zk)
    if pq == pk and nq == nk:
        queries = keys
    elif pk <= pq < pk + Zk.numel() * 4 and (pq - pk) % (P * D * 4) == 0 and pq + Zq.numel() * 4 <= pk + Zk.numel() * 4:
        first = (pq - pk) // (P * D * 4)                      # the query shard is a slice of the key set
        queries = keys.slice(first, first + nq)
    else:
        queries = pack_images(Zq)
    mean = reduce == REDUCE_MEAN_OTHERS
    shift = fixed_shift(max_sq([keys] if queries is keys else [keys, queries], Zq.device), nk) if mean else None
    acc = new_acc(nq * P, reduce, Zq.device)
    chunk = max(1, min(nq, minbuf_budget // (nk * P * 4)))
    for q0 in range(0, nq, chunk):
        qn = min(chunk, nq - q0)
        qs = queries if (qn == nq and queries is keys) else queries.slice(q0, q0 + qn)
        mb = pair_block(keys, qs)
        pair_accumulate(mb, P, reduce, self_offset + q0, shift, acc[q0 * P:(q0 + qn) * P])
    inside = 0 <= self_offset and self_offset + nq <= nk
    return pair_finalize(acc, P, reduce, nk - 1 if (mean and inside) else nk, shift)


def pair_min_sq_part(Z, part=0, nparts=1):
    """One slice of the SYMMETRIC all-pairs problem on Z [N, P, D] (keys == queries): returns
    minbuf fp32 [N, N*P] with minbuf[j][i*P + p] = min_q ||Z[i][p] - Z[j][q]||^2 for the (i, j) tile
    pairs of this slice and +inf elsewhere.  The element-wise minimum over all `nparts` slices (e.g.
    a MIN all-reduce across ranks) is the complete matrix; `pair_reduce` turns it into w."""
    pk = pack_images(Z)
    return pair_block(pk, pk, part=part, nparts=nparts).view(torch.float32)


def pair_reduce(minbuf, P, reduce, self_first=0, max_sq_t=None, n_keys=None):
    """minbuf fp32 [nk, nq*P] (squared mins) -> w fp32 [nq, P]: mean over key images != self_first + i
    of the square roots, or the square root of the min over all key images.  The fixed-point scale of the
    mean comes from max_sq_t (fp32 [1], the category's largest squared norm); without it the bound is taken
    from the minbuf itself (d <= sqrt(max d2))."""
    nk, cols = minbuf.shape
    nq = cols // P
    if nq == 0:                      # a rank with an empty image shard
        return torch.empty((0, P), dtype=torch.float32, device=minbuf.device)
    bits = minbuf.view(torch.int32)
    shift = None
    if reduce == REDUCE_MEAN_OTHERS:
        if max_sq_t is None:         # d <= sqrt(max d2) = 2 sqrt(max d2 / 4)
            max_sq_t = (minbuf.view(torch.float32).max().reshape(1) / 4).contiguous()
        shift = fixed_shift(max_sq_t, n_keys if n_keys is not None else nk)
    acc = new_acc(cols, reduce, minbuf.device)
    pair_accumulate(bits, P, reduce, self_first, shift, acc)
    inside = 0 <= self_first and self_first + nq <= nk
    return pair_finalize(acc, P, reduce, nk - 1 if (reduce == REDUCE_MEAN_OTHERS and inside) else nk, shift)


# ---- supervised weights: pruned exact nearest neighbour (csrc/alpha.hip)
NN_CAP = 64      # candidate slots per query patch and key set; a longer list sends that query to the exhaustive kernel


def h16_from_packed(pk):
    """The hi plane of a Packed block as plain fp16 rows: uint8 [n*P, D*2]."""
    out = torch.empty((pk.n * pk.P, pk.D * 2), dtype=torch.uint8, device=pk.hl.device)
    if pk.n:
        _check(lib().ac_h16_from_hl32(pk.hl.data_ptr(), pk.n * pk.P, pk.D, out.data_ptr(), _stream()))
    return out


def nn_slack(max_sq_keys, max_sq_queries):
    """fp32 [3] on the device: the candidate window (relative, absolute) and the fp16-range flag (include/ac_hip.h)."""
    s = torch.empty(3, dtype=torch.float32, device=max_sq_keys.device)
    _check(lib().ac_nn_slack(_dev(max_sq_keys, torch.float32, "max_sq_keys"), _dev(max_sq_queries, torch.float32, "max_sq_queries"),
                             s.data_ptr(), _stream()))
    return s


class NNState:
    """Running state of one pruned search: per query row the approximate minimum, the exact minimum so far, the
    candidate list of the key set being scanned, and - sticky over the key sets - which queries had a list longer
    than their slots (overflow_rows) and whether any had (overflow: the ONE word the host reads, once per search)."""

    def __init__(self, nq_rows, device, cap=NN_CAP, overflow=None):
        self.n, self.cap = nq_rows, cap
        self.approx_min = filled((nq_rows,), torch.int32, device, INF_BITS)
        self.best = filled((nq_rows,), torch.float64, device, F64_INF_BITS)
        self.cand = torch.empty((nq_rows, cap), dtype=torch.int32, device=device)
        self.count = filled((nq_rows,), torch.int32, device, 0)
        self.overflow_rows = filled((nq_rows,), torch.int32, device, 0)
        # `overflow`: a 1-element int32 view of the caller's flag vector (several searches, ONE all-reduce, ONE read)
        self.overflow = filled((1,), torch.int32, device, 0) if overflow is None else fill_(overflow, 0)


def nn_scan(state, keys, keys_h16, queries, queries_h16, slack):
    """One key set against all queries: pass 1 (candidates) + pass 2 (exact distances of the candidates).  Nothing is
    read back: a query whose list overflowed is marked in state.overflow_rows for nn_fix_overflow at the end."""
    if keys.n == 0 or queries.n == 0:
        return
    nk, nq, D = keys.n * keys.P, queries.n * queries.P, keys.D
    fill_(state.count, 0)
    ev0, ev1 = _event_pair()
    _check(lib().ac_nn_candidates(keys_h16.data_ptr(), keys.sq.data_ptr(), nk, queries_h16.data_ptr(), queries.sq.data_ptr(), nq, D,
                                  slack.data_ptr(), state.approx_min.data_ptr(), state.cand.data_ptr(), state.count.data_ptr(),
                                  state.cap, 0, _stream()))
    if ev0 is not None:
        ev1.record()
        kernel_events.append(("nn_candidates", ev0, ev1, 1.0 * keys.n * queries.n, keys.P, D))
    _check(lib().ac_nn_refine(keys.hl.data_ptr(), nk, queries.hl.data_ptr(), nq, D, state.cand.data_ptr(), state.count.data_ptr(),
                              state.cap, 0, state.best.data_ptr(), state.overflow.data_ptr(), state.overflow_rows.data_ptr(),
                              slack.data_ptr(), _stream()))


class OverflowQueries:
    """The overflowed query rows of a search as a compact Packed block (padded to whole images) + their row indices
    + the uint32 min accumulator the exhaustive blocks fold into."""

    def __init__(self, state, queries):
        dev = queries.hl.device
        P, D = queries.P, queries.D
        self.idx = torch.empty((max(state.n, 1),), dtype=torch.int32, device=dev)
        self.n_rows = 0
        self.packed = None
        if state.n == 0:
            # a rank without query images (fewer images than ranks) enters here whenever ANY rank overflowed (the flag is
            # MAX all-reduced): it has nothing to re-measure - its zero-size tensors have no address to hand to the
            # library - but still carries the bank shards round the second ring (scan / commit are no-ops)
            return
        n_out = torch.empty(1, dtype=torch.int32, device=dev)
        _check(lib().ac_nn_overflow_compact(state.overflow_rows.data_ptr(), state.n, self.idx.data_ptr(), n_out.data_ptr(), _stream()))
        self.n_rows = int(n_out.item())                        # slow path: the host sizes the compact copy
        if self.n_rows:
            rows = -(-self.n_rows // P) * P                    # the kernel's query axis comes in whole "images" of P rows
            hl = torch.empty((rows, D * 4), dtype=torch.uint8, device=dev)
            sq = torch.empty((rows,), dtype=torch.float32, device=dev)
            _check(lib().ac_gather_rows(queries.hl.data_ptr(), D * 4, self.idx.data_ptr(), self.n_rows, rows, hl.data_ptr(), _stream()))
            _check(lib().ac_gather_rows(queries.sq.data_ptr(), 4, self.idx.data_ptr(), self.n_rows, rows, sq.data_ptr(), _stream()))
            self.packed = Packed(hl, sq, rows // P, P, D)
            self.acc = new_acc(rows, REDUCE_MIN_ALL, dev)

    def scan(self, keys, budget=1 << 30):
        """Exhaustive one-way split-fp16 blocks: one key set against the compact queries."""
        if not self.n_rows or keys is None or keys.n == 0:
            return
        Qc = self.packed
        c = max(1, int(budget // (4 * Qc.P * max(Qc.n, 1))))   # key images per launch: minbuf [c, Qc.n * P] in the budget
        for k0 in range(0, keys.n, c):
            mb = pair_block(keys.slice(k0, min(keys.n, k0 + c)), Qc)
            pair_accumulate(mb, Qc.P, REDUCE_MIN_ALL, NO_SELF, None, self.acc)

    def commit(self, state):
        if self.n_rows:
            _check(lib().ac_nn_scatter_exhaustive(self.acc.data_ptr(), self.idx.data_ptr(), self.n_rows, state.best.data_ptr(), _stream()))


def nn_any_overflow(state):
    """The search's one host read: did any query overflow on any key set (or the norms leave the fp16 range)?"""
    return bool(int(state.overflow.item()))


def nn_finalize(state, P):
    w = torch.empty((state.n // P, P), dtype=torch.float32, device=state.best.device)
    if state.n:
        _check(lib().ac_nn_finalize(state.best.data_ptr(), state.n, w.data_ptr(), _stream()))
    return w


def alpha_softmax(w, tau, k=1.0, want_f32=False):
    """w fp32 [N, P] -> alpha fp64 [N, P] (utils.py:246-254); want_f32: also its fp32 rounding (main.py:294's
    `.float()`), written by the same kernel -> (alpha, alpha_f32)."""
    p = _dev(w, torch.float32, "w")
    N, P = w.shape
    a = torch.empty((N, P), dtype=torch.float64, device=w.device)
    a32 = torch.empty((N, P), dtype=torch.float32, device=w.device) if want_f32 else None
    if N:
        _check(lib().ac_alpha_softmax_f32(p, N, P, float(tau), float(k), a.data_ptr(), a32.data_ptr() if want_f32 else None, _stream()))
    return (a, a32) if want_f32 else a


def alpha_average(n, P, device):
    """The `average` weights (examples/main.py:290-291: ones / P) as fp32 [n, P]: the bit pattern of float32(1) /
    float32(P) - what torch's `ones(...) / P` evaluates to - filled by the library."""
    import struct
    import numpy as np
    bits = struct.unpack("<I", struct.pack("<f", float(np.float32(1.0) / np.float32(P))))[0]
    return filled((n, P), torch.float32, device, bits)


def weighted_avg(alpha, Z, out=None):
    """alpha fp32 [N, P] (or [N,1,P]), Z fp32 [N, P, D] -> X fp32 [N, D] (out: contiguous rows to write into)."""
    N, P, D = Z.shape
    if out is not None and (tuple(out.shape) != (N, D) or out.dtype != torch.float32 or not out.is_contiguous()):
        raise ValueError("weighted_avg: out must be contiguous fp32 [%d, %d]" % (N, D))
    if N == 0:                           # a rank with an empty image shard
        return out if out is not None else torch.empty((0, D), dtype=torch.float32, device=Z.device)
    alpha = alpha.reshape(alpha.shape[0], -1).contiguous()
    pa, pz = _dev(alpha, torch.float32, "alpha"), _dev(Z, torch.float32, "Z")
    if alpha.shape != (N, P):
        raise ValueError("alpha %s does not match Z %s" % (tuple(alpha.shape), tuple(Z.shape)))
    X = torch.empty((N, D), dtype=torch.float32, device=Z.device) if out is None else out
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


# ---------------------------------------------------------------- clustering
def ward_linkage(condensed, n):
    """scipy.cluster.hierarchy.linkage(y, "ward") on the device: condensed fp64 distances (as pdist() returns
    them) -> Z fp64 [n-1, 4], bit-identical to scipy's."""
    p = _dev(condensed, torch.float64, "condensed")
    if condensed.numel() != n * (n - 1) // 2:
        raise ValueError("condensed has %d entries, n = %d needs %d" % (condensed.numel(), n, n * (n - 1) // 2))
    Z = torch.empty((max(n - 1, 0), 4), dtype=torch.float64, device=condensed.device)
    if n < 2:
        return Z
    nbytes = lib().ac_ward_workspace_bytes(n)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=condensed.device)      # ~8 n^2 bytes: not kept around
    bad = torch.empty(1, dtype=torch.int32, device=condensed.device)
    _check(lib().ac_ward_linkage(p, n, Z.data_ptr(), bad.data_ptr(), ws.data_ptr(), nbytes, _stream()))
    flag = int(bad.item())
    if flag == 1:
        raise ValueError("The condensed distance matrix must contain only finite values.")     # scipy's message
    if flag:
        raise ValueError("Ward linkage: the Lance-Williams update produced NaN distances (a cluster was left without a "
                         "comparable neighbour); the input is numerically degenerate, e.g. exactly collinear points")
    return Z


def cluster_scores(labels_true, labels_pred, k_true, k_pred):
    """int32 device label vectors -> (NMI, ARI, contingency int64 [k_true, k_pred] on the host)."""
    a, b = _dev(labels_true, torch.int32, "labels_true"), _dev(labels_pred, torch.int32, "labels_pred")
    n = labels_true.numel()
    if labels_pred.numel() != n or n == 0:
        raise ValueError("label vectors must be non-empty and of equal length")
    C = torch.empty((k_true, k_pred), dtype=torch.int64, device=labels_true.device)
    scores = torch.empty(2, dtype=torch.float64, device=labels_true.device)
    bad = torch.empty(1, dtype=torch.int32, device=labels_true.device)
    _check(lib().ac_cluster_scores(a, b, n, k_true, k_pred, C.data_ptr(), scores.data_ptr(), bad.data_ptr(), _stream()))
    if int(bad.item()):
        raise ValueError("a label lies outside [0, k)")
    s = scores.cpu()
    return float(s[0]), float(s[1]), C.cpu().numpy()
