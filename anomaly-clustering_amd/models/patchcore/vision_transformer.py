"""DINO ViT whose forward runs on hand-written HIP kernels (gfx950).

Drop-in for the reference's ``patchcore.vision_transformer`` on the hot path
(reference: models/patchcore/vision_transformer.py:134-255): same constructor
names (``vit_tiny / vit_small / vit_base``), the same module tree and
parameter names as a DINO ``state_dict`` (``patch_embed.proj``, ``cls_token``,
``pos_embed``, ``blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}``,
``norm``) and the same seeded initialisation, so ``load_state_dict`` of a DINO
checkpoint and ``torch.manual_seed(s)``-reproducible random weights both work.

The arithmetic is NOT torch: ``Block.forward`` and ``prepare_tokens`` call the
C-ABI library (include/ac_hip.h) through ``acnative``; the torch modules here
only own the parameters.  Blocks stay ``nn.Module``s that are *called*, so
forward hooks (NetworkFeatureAggregator) see each block's fp32 output exactly
as in the reference.  There is no CPU fallback: without the HIP library, or
on a CPU tensor, forward raises.

Out of scope: DINOHead, get_last_selfattention,
the bicubic pos-embed resize branch (inputs must match the native grid).
"""
import math
from functools import partial

import torch
import torch.nn as nn

import acnative


class Mlp(nn.Module):
    """Parameter holder for fc1 -> exact GELU -> fc2 (reference :49-65)."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    """Parameter holder for the fused-QKV multi-head attention (reference :68-92)."""

    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    """One pre-norm transformer block; forward = one native call (reference :95-113)."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self._packed = None

    def packed(self):
        """Device-resident split-fp16 copies of the four weight matrices."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, acnative.pack_block_weights(self))
        return self._packed[1]

    def forward(self, x):
        return acnative.vit_block(self, x)


class PatchEmbed(nn.Module):
    """Conv(k=s=patch) patch projection; the conv itself runs as an im2col GEMM."""

    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=(224,), patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.patch_embed = PatchEmbed(img_size[0], patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList(
            [Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # timm's ViTs (backbones.py:35-37 of the reference) carry a classification head; the clustering path stops at a
        # tapped block, so the head only holds its checkpoint tensors (strict loading) and is never run
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self._packed = None
        self.reset_parameters()

    def reset_parameters(self):
        """Same draws, in the same order, as the reference initialiser (:161-172):
        pos_embed, cls_token, then every Linear weight in module order; Linear
        biases 0, LayerNorm (1, 0).  The conv keeps torch's default init."""
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.zeros_(m.bias)
                nn.init.ones_(m.weight)

    def packed(self):
        ps = (self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.cls_token, self.pos_embed)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, acnative.pack_embed_weights(self))
        return self._packed[1]

    def prepare_tokens(self, x):
        B, nc, w, h = x.shape
        ps = self.patch_embed.patch_size
        if (w // ps) * (h // ps) + 1 != self.pos_embed.shape[1] or w != h:
            raise ValueError("input %dx%d does not match the native token grid (pos-embed resize is out of scope)"
                             % (w, h))
        return acnative.vit_prepare_tokens(self, x)

    def forward(self, x):
        x = self.prepare_tokens(x)
        for blk in self.blocks:
            x = blk(x)
        if not isinstance(self.head, nn.Identity):
            raise NotImplementedError("the classification head is outside the anomaly-clustering path (tap a block)")
        return acnative.layernorm_rows(x[:, 0].contiguous(), self.norm)

    def get_intermediate_layers(self, x, n=1):
        x = self.prepare_tokens(x)
        out = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if len(self.blocks) - i <= n:
                B, T, C = x.shape
                out.append(acnative.layernorm_rows(x.reshape(B * T, C), self.norm).reshape(B, T, C))
        return out


def _vit(embed_dim, num_heads, patch_size, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=embed_dim, depth=kwargs.pop("depth", 12),
                             num_heads=num_heads, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_tiny(patch_size=16, **kwargs):
    return _vit(192, 3, patch_size, **kwargs)


def vit_small(patch_size=16, **kwargs):
    return _vit(384, 6, patch_size, **kwargs)


def vit_base(patch_size=16, **kwargs):
    return _vit(768, 12, patch_size, **kwargs)
