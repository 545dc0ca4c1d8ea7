"""Deterministic, RNG-free synthetic tensors shared by the golden generator and
the tests.  Values come from exact integer arithmetic, so the same array is
rebuilt bit-for-bit on any machine / numpy / torch version (the golden files
store only the *outputs* the reference produced for these inputs)."""
import numpy as np


def hash_tensor(shape, salt=0, scale=1.0, dtype=np.float32):
    """Pseudo-random values in [-scale/2, scale/2) from a 64-bit integer mix of
    the flat index (splitmix-style multiply/xor-shift, all in uint64)."""
    n = int(np.prod(shape))
    x = np.arange(n, dtype=np.uint64) + np.uint64((salt * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)   # 24-bit mantissa: exact in fp32
    return ((u - 0.5) * scale).astype(dtype).reshape(shape)


def gaussish_tensor(shape, salt=0, scale=1.0, dtype=np.float32):
    """Sum of four hash uniforms: bell-shaped, std ~= scale * 0.577."""
    acc = np.zeros(shape, dtype=np.float64)
    for t in range(4):
        acc += hash_tensor(shape, salt * 4 + t, 1.0, np.float64)
    return (acc * scale).astype(dtype)


def clustered_embeddings(n_images, patches, dim, n_classes, salt=0):
    """Z-like tensor [N, P, D]: a shared 'normal' texture plus, per class, a few
    patches displaced along a class direction - gives alpha a real signal and
    the clustering a non-degenerate answer."""
    base = gaussish_tensor((1, patches, dim), salt + 1, 1.0, np.float64)
    noise = gaussish_tensor((n_images, patches, dim), salt + 2, 0.35, np.float64)
    dirs = gaussish_tensor((n_classes, dim), salt + 3, 2.5, np.float64)
    Z = base + noise
    labels = np.arange(n_images) % n_classes
    for i in range(n_images):
        c = labels[i]
        if c == 0:
            continue
        p0 = (7 * c + 3 * i) % max(1, patches - 4)
        Z[i, p0:p0 + 4] += dirs[c]
    return Z.astype(np.float32), labels


def dino_like_statistics(sd):
    """In place: give a seed-0 ViT state_dict (dict of torch tensors) the activation statistics of a TRAINED
    DINO ViT, which the trunc-normal(0.02) init lacks - all from the RNG-free hashes above:
      * 6 outlier channels of the residual stream (x50: patch projection rows, their bias and pos_embed columns) -
        after LayerNorm the ordinary channels shrink by ~10x, which is what stresses the low half of the
        split-fp16 operands;
      * LayerNorm gains log-uniform in [0.2, 5], LayerNorm biases and every linear bias non-zero;
      * heavy-tailed fc1 / fc2 / proj weights: one element in 1000 multiplied by 20.
    """
    import torch
    C = sd["pos_embed"].shape[-1]
    outliers = [7, 130, 301, 444, 589, 702][: max(1, min(6, C // 64))]
    outliers = [c % C for c in outliers]
    w = sd["patch_embed.proj.weight"]
    for c in outliers:
        w[c] *= 50.0
        sd["pos_embed"][..., c] *= 50.0
        sd["cls_token"][..., c] = 1.0 + 0.01 * c
    sd["patch_embed.proj.bias"].copy_(torch.from_numpy(hash_tensor((C,), 901, 0.4)))
    salt = 1000
    for name in sorted(sd):
        t = sd[name]
        salt += 1
        if name.endswith(("norm1.weight", "norm2.weight")):
            u = torch.from_numpy(hash_tensor(tuple(t.shape), salt, 1.0)).double() + 0.5          # [0, 1)
            t.copy_(torch.exp(u * (np.log(5.0) - np.log(0.2)) + np.log(0.2)).float())
        elif name.endswith(("norm1.bias", "norm2.bias")):
            t.copy_(torch.from_numpy(hash_tensor(tuple(t.shape), salt, 1.0)))
        elif name.startswith("blocks.") and name.endswith(".bias"):
            t.copy_(torch.from_numpy(hash_tensor(tuple(t.shape), salt, 0.6)))
        elif name.endswith(("fc1.weight", "fc2.weight", "proj.weight")) and name.startswith("blocks."):
            pick = torch.from_numpy(hash_tensor(tuple(t.shape), salt, 1.0)) > 0.499              # ~1 in 1000
            t[pick] *= 20.0
    return sd
