"""Oracle: DINO ViT forward up to (and including) a given block, on the CPU.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Restates models/patchcore/vision_transformer.py as a function of a DINO-format
``state_dict`` (no nn.Module, no hooks): the tap semantics of
models/patchcore/common.py:259-288 (output of ``blocks.i`` is taken straight
after the block, before the final ``norm``; the forward stops after the last
requested block) are folded into ``taps``.
"""
import math

import torch
import torch.nn.functional as F


def geometry(sd):
    """Infer (embed_dim, depth, patch_size, tokens) from a state_dict."""
    w = sd["patch_embed.proj.weight"]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    return w.shape[0], depth, w.shape[-1], sd["pos_embed"].shape[1]


def prepare_tokens(sd, images):
    """vision_transformer.py:196-207 (+ PatchEmbed :128-131).

    conv 8x8/8 -> flatten(2).transpose(1,2) -> cat CLS -> + pos_embed.
    The interpolate_pos_encoding resize branch (:175-194) is out of scope: the
    path always runs at the native 224x224 grid where it returns pos_embed.
    """
    x = F.conv2d(images, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"],
                 stride=sd["patch_embed.proj.weight"].shape[-1])
    x = x.flatten(2).transpose(1, 2)
    cls = sd["cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    assert x.shape[1] == sd["pos_embed"].shape[1], "oracle covers the native grid only"
    return x + sd["pos_embed"]


def attention(sd, pre, x, num_heads):
    """vision_transformer.py:80-92 (qkv with bias, scale = head_dim**-0.5)."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])


def mlp(sd, pre, x):
    """vision_transformer.py:59-65 (exact-erf GELU: nn.GELU default)."""
    h = F.linear(x, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    h = F.gelu(h)
    return F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])


def block(sd, i, x, num_heads, eps=1e-6):
    """vision_transformer.py:107-113; LayerNorm eps 1e-6 from :253 (vit_base)."""
    pre = "blocks.%d." % i
    C = x.shape[-1]
    y = attention(sd, pre, F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps),
                  num_heads)
    x = x + y
    z = mlp(sd, pre, F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps))
    return x + z


@torch.no_grad()
def forward_taps(sd, images, taps=(10, 11), num_heads=12, dtype=torch.float32):
    """Run the ViT and return {block_index: [B, T, C]} for each tap (fp32 = the reference's
    arithmetic; dtype=torch.float64 gives the "truth" that stress tests compare both sides to).

    Equivalent to NetworkFeatureAggregator.forward (common.py:259-268) with
    layers_to_extract_from = ["blocks.<i>" for i in taps].
    """
    sd = {k: v.detach().to(dtype).cpu() for k, v in sd.items()}
    x = prepare_tokens(sd, images.to(dtype).cpu())
    out = {}
    last = max(taps)
    for i in range(last + 1):
        x = block(sd, i, x, num_heads)
        if i in taps:
            out[i] = x.clone()
    return out


def flops_per_image(embed_dim=768, depth=12, tokens=785, patch_pixels=192, mlp_ratio=4):
    """Dense-contraction FLOPs of one image (156.3 GFLOP)."""
    C, T = embed_dim, tokens
    per_block = 2 * T * C * 3 * C + 2 * 2 * T * T * C + 2 * T * C * C + 2 * 2 * T * C * mlp_ratio * C
    return depth * per_block + 2 * (T - 1) * patch_pixels * C
