"""Backbone registry (reference: models/patchcore/backbones.py:53-79).

On the ViT hot path the DINO names exist; as in the reference, every DINO name builds ``vit_base`` with the
name's patch size (backbones.py:70).  The reference downloads the checkpoint from dl.fbaipublicfiles.com;
here (no network) the weights come from

  1. ``$AC_DINO_WEIGHTS/<checkpoint file name>`` or the file ``$AC_DINO_WEIGHTS`` itself (a DINO-format
     ``state_dict`` saved with torch.save), loaded with ``strict=True``.  If the variable is set and the
     file is not there, load() RAISES - it never silently substitutes other weights;
  2. or, only when the caller opts in with ``$AC_BACKBONE_SEED=<int>``, the seeded reference initialisation
     (trunc-normal 0.02 etc.: the benchmark / parity configuration).  The model then carries
     ``weights_source = "random-init(seed=N)"``, a warning is printed, and examples/main.py writes that
     string next to its artefacts.
  With neither, load() raises: clustering with random weights must never look like a real run.

``wideresnet50`` (BASELINE config 1, reference backbones.py:27: ``models.wide_resnet50_2(pretrained=True)``) is
patchcore/wide_resnet.py: torchvision's published architecture under torchvision's module names, its forward on
the MI355X (convolutions as split-fp16 MFMA GEMMs, csrc/cnn.hip).  Weights: ``$AC_WRN50_WEIGHTS`` (torchvision's
``wide_resnet50_2-*.pth`` state_dict, file or directory, ``strict=True``), or - opt-in, as above - torchvision's own
initialisation from ``$AC_BACKBONE_SEED``.

``vit_small`` / ``vit_base`` / ``vit_large`` / ``vit_deit_base`` (reference backbones.py:34-38: timm's patch-8 ViTs) are
the same transformer with a classification head that the path never reaches; their timm ``state_dict`` (key names
identical to DINO's plus ``head.*``) loads strictly from ``$AC_TIMM_WEIGHTS/<timm model name>.pth``.  The other
torchvision / timm names (CNN families, Swin, hybrid ViTs) raise.
"""
import os
import sys

import torch

import patchcore.vision_transformer as vits

_DINO = {
    "dino_deitsmall16": ("dino_deitsmall16_pretrain.pth", 16),
    "dino_deitsmall8_300ep": ("dino_deitsmall8_300ep_pretrain.pth", 8),
    "dino_vitbase16": ("dino_vitbase16_pretrain.pth", 16),
    "dino_vitbase8": ("dino_vitbase8_pretrain.pth", 8),
}


# timm ViTs of the reference's registry (backbones.py:34-38: timm.create_model("<model>", pretrained=True)) whose
# state_dict has the key names of patchcore/vision_transformer.py (DINO's ViT is timm's): (timm model, patch, width,
# depth, heads).  Weights: $AC_TIMM_WEIGHTS/<timm model>.pth (a state_dict), strict.
_TIMM_VIT = {
    "vit_small": ("vit_small_patch8_224", 8, 384, 12, 6),
    "vit_base": ("vit_base_patch8_224", 8, 768, 12, 12),
    "vit_large": ("vit_large_patch8_224", 8, 1024, 24, 16),
    "vit_deit_base": ("deit_base_patch8_224", 8, 768, 12, 12),
}


def _find_weights(filename, var="AC_DINO_WEIGHTS"):
    root = os.environ.get(var)
    if not root:
        return None
    if os.path.isfile(root):
        return root
    cand = os.path.join(root, filename)
    return cand if os.path.isfile(cand) else None


def _seed_opt_in(name):
    seed = os.environ.get("AC_BACKBONE_SEED")
    if seed is None:
        raise RuntimeError(
            "no weights for backbone %r: set AC_DINO_WEIGHTS (DINO names) / AC_TIMM_WEIGHTS (vit_*) / AC_WRN50_WEIGHTS "
            "(wideresnet50) to a "
            "checkpoint (file or directory), or opt in "
            "to the seeded random initialisation with AC_BACKBONE_SEED=<int> (benchmarks / parity tests only)" % name)
    return int(seed)


_WRN50_FILE = "wide_resnet50_2-95faca4d.pth"      # torchvision's IMAGENET1K_V1 checkpoint (pretrained=True)


def _load_wrn50(name):
    from patchcore import wide_resnet
    root = os.environ.get("AC_WRN50_WEIGHTS")
    path = None
    if root:
        path = root if os.path.isfile(root) else os.path.join(root, _WRN50_FILE)
        if not os.path.isfile(path):
            raise FileNotFoundError("AC_WRN50_WEIGHTS=%r is set but %r was not found there" % (root, _WRN50_FILE))
    source = path
    if path is None:
        seed = _seed_opt_in(name)
        torch.manual_seed(seed)
        source = "random-init(seed=%d)" % seed
        print("WARNING: backbone %r runs on SEEDED RANDOM weights (%s), not torchvision's checkpoint" % (name, source),
              file=sys.stderr)
    model = wide_resnet.wide_resnet50_2()
    for p in model.parameters():
        p.requires_grad = False
    model.eval()
    if path is not None:
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True), strict=True)
    model.weights_source = source
    if torch.cuda.is_available():
        model.to(torch.device("cuda"))
    return model


def load(name):
    if name == "wideresnet50":
        return _load_wrn50(name)
    if name in _TIMM_VIT:
        timm_name, patch_size, width, depth, heads = _TIMM_VIT[name]
        return _load_vit(name, timm_name + ".pth", "AC_TIMM_WEIGHTS",
                         lambda: vits._vit(width, heads, patch_size, depth=depth, num_classes=1000))
    if name not in _DINO:
        raise NotImplementedError(
            "backbone %r is outside the MI355X hot path (only %s + wideresnet50); the other torchvision/timm "
            "backbones need those packages and a weight download" % (name, sorted(_DINO) + sorted(_TIMM_VIT)))
    filename, patch_size = _DINO[name]
    return _load_vit(name, filename, "AC_DINO_WEIGHTS", lambda: vits.vit_base(patch_size=patch_size, num_classes=0))


def _load_vit(name, filename, var, build):
    path = _find_weights(filename, var)
    if path is None and os.environ.get(var):
        raise FileNotFoundError("%s=%r is set but %r was not found there" % (var, os.environ[var], filename))
    source = path
    if path is None:
        seed = _seed_opt_in(name)
        torch.manual_seed(seed)
        source = "random-init(seed=%d)" % seed
        print("WARNING: backbone %r runs on SEEDED RANDOM weights (%s), not the pretrained checkpoint" % (name, source),
              file=sys.stderr)
    model = build()
    for p in model.parameters():
        p.requires_grad = False
    model.eval()
    if path is not None:
        state = torch.load(path, map_location="cpu", weights_only=True)
        model.load_state_dict(state, strict=True)
    model.weights_source = source
    if torch.cuda.is_available():
        model.to(torch.device("cuda"))
    return model
