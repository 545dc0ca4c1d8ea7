"""Image transform of the MVTec dataset: Resize(256) -> CenterCrop(224) -> ToTensor ->
Normalize, reference models/patchcore/datasets/mvtec.py:74-81.

What is pinned against what (torchvision is absent from this image):
  * torchvision's Resize / CenterCrop on a PIL image ARE Pillow's Image.resize(BILINEAR) / Image.crop with the
    sizes and offsets restated in patchcore/datasets/mvtec.py (documented equivalence; torchvision 0.13's
    functional_pil.resize -> img.resize, center_crop -> int(round((dim - crop) / 2.0))).
  * CPU: oracle/preprocess.py (an integer restatement of Pillow's two-pass resampling) == Pillow itself, bit for
    bit, on up- and down-scaling, both orientations and MVTec's sizes; == the product's host transform
    (MVTecDataset.transform_img); and == the committed fixture (tests/golden/a15_*.{png,npz}, written by
    make_golden.py --a15 from Pillow).
  * GPU: acnative.preprocess_images (csrc/preprocess.hip) == all of the above, bit for bit, and
    MVTecDataset.load_category (one pass, batched, on the device) == the per-item host path.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

SIZES = [(300, 320), (320, 300), (900, 900), (100, 90), (90, 100), (256, 256), (257, 300), (1024, 1024), (233, 401)]


def _pil_chain(arr):
    """The reference chain with Pillow + torch (what torchvision does for PIL inputs)."""
    import PIL.Image
    from patchcore.datasets import mvtec
    ds = mvtec.MVTecDataset.__new__(mvtec.MVTecDataset)
    ds.resize, ds.imagesize = 256, (3, 224, 224)
    ds._mean = torch.tensor(mvtec.IMAGENET_MEAN).view(3, 1, 1)
    ds._std = torch.tensor(mvtec.IMAGENET_STD).view(3, 1, 1)
    return ds.transform_img(PIL.Image.fromarray(arr)).numpy()


def _expected_from_fixture(g):
    """ToTensor + Normalize (torch, as the reference) of the fixture's resized + cropped uint8 pixels."""
    from patchcore.datasets import mvtec
    t = torch.from_numpy(g["cropped_u8"].copy()).to(torch.float32).div(255)
    t = (t - torch.tensor(mvtec.IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(mvtec.IMAGENET_STD).view(3, 1, 1)
    chk = t.double().flatten()
    np.testing.assert_allclose([chk.sum().item(), (chk * chk).sum().item(), chk[0].item(), chk[-1].item()], g["check"],
                               rtol=1e-12)
    return t.numpy()


def _random_image(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_oracle_resampling_equals_pillow():
    import PIL.Image
    from oracle import preprocess as op
    for i, (h, w) in enumerate(SIZES):
        img = _random_image(h, w, i)
        nw, nh = op.resized_size(w, h, 256)
        ref = np.asarray(PIL.Image.fromarray(img).resize((nw, nh), PIL.Image.BILINEAR))
        assert np.array_equal(op.resize_bilinear_u8(img, nw, nh), ref), (h, w)
        assert np.array_equal(op.transform_image(img), _pil_chain(img)), (h, w)


def test_committed_fixture():
    """tests/golden/a15_image.png through Pillow (make_golden.py --a15) -> tests/golden/a15_expected.npz."""
    import PIL.Image
    from oracle import preprocess as op
    g = np.load(os.path.join(GOLDEN, "a15_expected.npz"))
    img = np.asarray(PIL.Image.open(os.path.join(GOLDEN, "a15_image.png")).convert("RGB"))
    assert img.shape == tuple(g["shape"])
    out = op.transform_image(img)
    assert np.array_equal(out, _expected_from_fixture(g))
    assert np.array_equal(_pil_chain(img), out)


@pytest.mark.gpu
def test_gpu_transform_is_bit_identical(dev):
    import PIL.Image
    import acnative
    from oracle import preprocess as op
    for i, (h, w) in enumerate(SIZES):
        imgs = np.stack([_random_image(h, w, 10 * i + k) for k in range(3)])
        got = acnative.preprocess_images(torch.from_numpy(imgs).to(dev)).cpu().numpy()
        for k in range(3):
            assert np.array_equal(got[k], op.transform_image(imgs[k])), (h, w, k)
            assert np.array_equal(got[k], _pil_chain(imgs[k])), (h, w, k)
    img = np.asarray(PIL.Image.open(os.path.join(GOLDEN, "a15_image.png")).convert("RGB"))
    g = np.load(os.path.join(GOLDEN, "a15_expected.npz"))
    out = acnative.preprocess_images(torch.from_numpy(img[None].copy()).to(dev))[0].cpu().numpy()
    assert np.array_equal(out, _expected_from_fixture(g))
    with pytest.raises(RuntimeError, match="smaller than the crop"):
        acnative.preprocess_images(torch.zeros(1, 50, 400, 3, dtype=torch.uint8, device=dev), resize=100, crop=224)


@pytest.mark.gpu
def test_one_pass_category_loader_equals_per_item_host_path(dev, tmp_path):
    """MVTecDataset.load_category (decode once on the host, transform batched on the GPU, records collected on
    the way) against DataLoader(batch_size=1) over __getitem__ (the reference's two passes): same pixels, same
    records; images of two different sizes in one category."""
    import PIL.Image
    from patchcore.datasets import mvtec
    root = tmp_path / "mv"
    rng = np.random.default_rng(4)
    for split, kinds in (("train", ["good"]), ("test", ["good", "crack", "hole"])):
        for kind in kinds:
            os.makedirs(root / "nut" / split / kind)
            for i in range(3):
                hw = (300, 320) if (kind, i) != ("hole", 1) else (280, 290)
                PIL.Image.fromarray(rng.integers(0, 255, hw + (3,), dtype=np.uint8)).save(
                    root / "nut" / split / kind / ("%03d.png" % i))
    for kind in ("crack", "hole"):
        os.makedirs(root / "nut" / "ground_truth" / kind)
        for i in range(3):
            PIL.Image.fromarray(rng.integers(0, 2, (300, 320), dtype=np.uint8) * 255).save(
                root / "nut" / "ground_truth" / kind / ("%03d_mask.png" % i))
    ds = mvtec.MVTecDataset(str(root), "nut", split=mvtec.DatasetSplit.TEST)
    images, info = ds.load_category(dev, batch_size=4)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
    assert len(info) == len(ds) == 9 and images.shape == (9, 3, 224, 224)
    for i, item in enumerate(loader):
        assert torch.equal(images[i].cpu(), item["image"][0]), i
        del item["image"], item["mask"]
        assert set(info[i]) == set(item)
        for key in ("classname", "anomaly", "image_name", "image_path"):
            assert info[i][key] == item[key]
        assert torch.equal(info[i]["is_anomaly"], item["is_anomaly"])
    # a rank's view: only its shard (a range) / its bank images (a set) are decoded, the records are complete
    part, info_p = ds.load_category(dev, batch_size=2, only=(3, 7))
    assert torch.equal(part[3:7], images[3:7]) and [r["image_name"] for r in info_p] == [r["image_name"] for r in info]
    some, _ = ds.load_category(dev, only={0, 5, 8})
    assert all(torch.equal(some[i], images[i]) for i in (0, 5, 8))
