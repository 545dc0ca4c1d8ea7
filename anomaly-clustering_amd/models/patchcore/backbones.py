"""Backbone registry (reference: models/patchcore/backbones.py:53-79).

On the hot path only the DINO names exist.  As in the reference, every DINO name builds
``vit_base`` with the name's patch size (backbones.py:70).  The reference downloads the
checkpoint from dl.fbaipublicfiles.com; here weights come from, in this order:

  1. ``$AC_DINO_WEIGHTS/<checkpoint file name>`` or the file ``$AC_DINO_WEIGHTS`` itself
     (a DINO-format ``state_dict`` saved with torch.save), loaded with ``strict=True``;
  2. otherwise the seeded reference initialisation (``$AC_BACKBONE_SEED``, default 0) -
     the benchmark/parity configuration (no network on the build or GPU boxes).

The torchvision/timm names of the reference's table need those packages and a download;
asking for one raises with that explanation.
"""
import os

import torch

import patchcore.vision_transformer as vits

_DINO = {
    "dino_deitsmall16": ("dino_deitsmall16_pretrain.pth", 16),
    "dino_deitsmall8_300ep": ("dino_deitsmall8_300ep_pretrain.pth", 8),
    "dino_vitbase16": ("dino_vitbase16_pretrain.pth", 16),
    "dino_vitbase8": ("dino_vitbase8_pretrain.pth", 8),
}


def _find_weights(filename):
    root = os.environ.get("AC_DINO_WEIGHTS")
    if not root:
        return None
    if os.path.isfile(root):
        return root
    cand = os.path.join(root, filename)
    return cand if os.path.isfile(cand) else None


def load(name):
    if name not in _DINO:
        raise NotImplementedError(
            "backbone %r is outside the MI355X hot path (only %s); torchvision/timm backbones "
            "need those packages and a weight download" % (name, sorted(_DINO)))
    filename, patch_size = _DINO[name]
    path = _find_weights(filename)
    if path is None:
        torch.manual_seed(int(os.environ.get("AC_BACKBONE_SEED", "0")))
    model = vits.vit_base(patch_size=patch_size, num_classes=0)
    for p in model.parameters():
        p.requires_grad = False
    model.eval()
    if path is not None:
        state = torch.load(path, map_location="cpu", weights_only=True)
        model.load_state_dict(state, strict=True)
    if torch.cuda.is_available():
        model.to(torch.device("cuda"))
    return model
