"""Oracle: the image transform of the MVTec dataset, restated in integer arithmetic.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Reference: models/patchcore/datasets/mvtec.py:74-110 -
    transforms.Resize(256) -> CenterCrop(224) -> ToTensor() -> Normalize(ImageNet mean / std)
on a PIL RGB image.  torchvision is NOT in this image, so the pin is one level down: for PIL
inputs torchvision's Resize(int) calls PIL's ``Image.resize((w', h'), BILINEAR)`` with the short edge set to
the int and the long edge ``int(size * long / short)``; CenterCrop crops at ``int(round((dim - crop) / 2.0))``;
ToTensor is uint8 -> float32 / 255; Normalize is (x - mean) / std in float32.  This file restates Pillow's
resampling (src/libImaging/Resample.c, 8 bits per channel: two passes - horizontal then vertical - each with
float64 filter weights over a support scaled by the reduction factor, normalised, rounded to 22-bit fixed
point, accumulated in int32 from a rounding bias of 2^21, shifted and clamped to uint8) and is checked
against Pillow itself, bit for bit, in tests/test_preprocess.py; the HIP kernel (csrc/preprocess.hip) is
checked against both.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resized_size(w, h, resize):
    """torchvision Resize(int): short edge -> resize, long edge truncated."""
    if w <= h:
        return resize, int(resize * h / w)
    return int(resize * w / h), resize


def crop_offset(dim, crop):
    return int(round((dim - crop) / 2.0))


def bilinear_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear ("triangle") filter over the
    whole axis: bounds [out, 2] (first tap, tap count) and int32 weights [out, ksize]."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One resampling pass along `axis` (0 = vertical, 1 = horizontal) of a uint8 [H, W, C] image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        first, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[o, :n].astype(np.int64), src[first:first + n], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_w, out_h):
    """Pillow's Image.resize((out_w, out_h), BILINEAR) for a uint8 [H, W, C] array."""
    h, w = img.shape[:2]
    bh, kh = bilinear_coeffs(w, out_w)
    bv, kv = bilinear_coeffs(h, out_h)
    return _pass(_pass(img, bh, kh, 1), bv, kv, 0)          # horizontal first, uint8 in between


def transform_image(img, resize=256, crop=224, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 [H, W, 3] -> float32 [3, crop, crop]: the whole chain of mvtec.py:74-81."""
    h, w = img.shape[:2]
    nw, nh = resized_size(w, h, resize)
    r = resize_bilinear_u8(img, nw, nh)
    top, left = crop_offset(nh, crop), crop_offset(nw, crop)
    r = r[top:top + crop, left:left + crop]
    x = r.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(np.transpose(x, (2, 0, 1)))
