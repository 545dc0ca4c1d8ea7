"""MVTec-AD folder dataset without torchvision (reference: models/patchcore/datasets/mvtec.py:30-167).

Same constructor, same sorted folder walk, same item dict.  The torchvision transform chain
Resize(resize) -> CenterCrop(imagesize) -> ToTensor -> Normalize(ImageNet) is restated with PIL +
torch: torchvision's Resize on a PIL image is ``Image.resize`` (bilinear) with the short edge set
to `resize`, CenterCrop rounds the offsets, ToTensor scales uint8 to [0, 1].

When `source` does not exist on disk (the build and GPU boxes have no dataset) the caller should
use ``patchcore.datasets.synthetic.SyntheticMVTecDataset``; ``MVTecDataset`` raises a clear error.
"""
import os
from enum import Enum

import numpy as np
import torch

_CLASSNAMES = [
    "bottle", "cable", "capsule", "carpet", "grid", "hazelnut", "leather", "metal_nut", "pill", "screw",
    "tile", "toothbrush", "transistor", "wood", "zipper",
]

IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]


class DatasetSplit(Enum):
    TRAIN = "train"
    VAL = "val"
    TEST = "test"


def _resize_center_crop(img, resize, imagesize):
    import PIL.Image
    w, h = img.size
    if w <= h:
        nw, nh = resize, int(resize * h / w)
    else:
        nh, nw = resize, int(resize * w / h)
    img = img.resize((nw, nh), PIL.Image.BILINEAR)
    top = int(round((nh - imagesize) / 2.0))
    left = int(round((nw - imagesize) / 2.0))
    return img.crop((left, top, left + imagesize, top + imagesize))


def _to_tensor(img):
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.array(arr)).permute(2, 0, 1)       # a writable copy (PIL hands out a read-only view)
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


class MVTecDataset(torch.utils.data.Dataset):
    def __init__(self, source, classname, resize=256, imagesize=224, split=DatasetSplit.TRAIN,
                 train_val_split=1.0, **kwargs):
        super().__init__()
        self.source = source
        self.split = split
        self.classnames_to_use = [classname] if classname is not None else _CLASSNAMES
        self.train_val_split = train_val_split
        self.resize = resize
        self.imagesize = (3, imagesize, imagesize)
        if not os.path.isdir(source):
            raise FileNotFoundError(
                "MVTec-AD root %r not found; use patchcore.datasets.synthetic.SyntheticMVTecDataset "
                "(examples/main.py --dataset synthetic) when no dataset is available" % (source,))
        self.imgpaths_per_class, self.data_to_iterate = self.get_image_data()
        self._mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
        self._std = torch.tensor(IMAGENET_STD).view(3, 1, 1)

    def transform_img(self, img):
        x = _to_tensor(_resize_center_crop(img, self.resize, self.imagesize[1]))
        return (x - self._mean) / self._std

    def transform_mask(self, img):
        return _to_tensor(_resize_center_crop(img, self.resize, self.imagesize[1]))

    def __getitem__(self, idx):
        import PIL.Image
        classname, anomaly, image_path, mask_path = self.data_to_iterate[idx]
        image = self.transform_img(PIL.Image.open(image_path).convert("RGB"))
        if self.split == DatasetSplit.TEST and mask_path is not None:
            mask = self.transform_mask(PIL.Image.open(mask_path))
        else:
            mask = torch.zeros([1, *image.size()[1:]])
        return {
            "image": image,
            "mask": mask,
            "classname": classname,
            "anomaly": anomaly,
            "is_anomaly": int(anomaly != "good"),
            "image_name": "/".join(image_path.split("/")[-4:]),
            "image_path": image_path,
        }

    def __len__(self):
        return len(self.data_to_iterate)

    def load_category(self, device, batch_size=64, only=None):
        """ONE pass over the split for the MI355X pipeline (the reference walks the dataset twice with batch size 1:
        once for the per-image records, examples/main.py:254-262, once to embed, :266): every file is decoded once
        on the host (PIL), the uint8 pixels of equally sized images go to the device in batches and the whole
        transform chain runs there (acnative.preprocess_images: bit-identical to transform_img).
        Returns (images fp32 [N, 3, S, S] on `device`, info) with info[i] = the item dict without pixels, in the
        form DataLoader(batch_size=1) collates it (what examples/test.py reads).
        only = (lo, hi) or a set of indices: decode and transform only those images - a rank of a multi-GPU run
        needs the pixels of its own shard (the other rows of `images` stay unwritten) but the records of all."""
        import PIL.Image
        import acnative
        n = len(self.data_to_iterate)
        size = self.imagesize[1]
        images = torch.empty((n, 3, size, size), dtype=torch.float32, device=device)
        info = []
        pending, shape = [], None

        def flush():
            if not pending:
                return
            idx = [i for i, _ in pending]
            u8 = torch.from_numpy(np.stack([a for _, a in pending])).to(device, non_blocking=True)
            out = acnative.preprocess_images(u8, self.resize, size, IMAGENET_MEAN, IMAGENET_STD)
            if idx == list(range(idx[0], idx[0] + len(idx))):
                images[idx[0]: idx[0] + len(idx)] = out
            else:
                images[torch.tensor(idx, device=device)] = out
            pending.clear()

        if only is None:
            wanted = range(n)
        elif isinstance(only, tuple):
            wanted = range(only[0], only[1])
        else:
            wanted = set(int(i) for i in only)
        for i, (classname, anomaly, image_path, _mask) in enumerate(self.data_to_iterate):
            if i in wanted:
                arr = np.asarray(PIL.Image.open(image_path).convert("RGB"))
                if shape is not None and (arr.shape != shape or len(pending) >= batch_size):
                    flush()
                shape = arr.shape
                pending.append((i, arr))
            info.append({"classname": [classname], "anomaly": [anomaly],
                         "is_anomaly": torch.tensor([int(anomaly != "good")]),
                         "image_name": ["/".join(image_path.split("/")[-4:])], "image_path": [image_path]})
        flush()
        return images, info

    def get_image_data(self):
        per_class, masks = {}, {}
        for classname in self.classnames_to_use:
            classpath = os.path.join(self.source, classname, self.split.value)
            maskpath = os.path.join(self.source, classname, "ground_truth")
            per_class[classname], masks[classname] = {}, {}
            for anomaly in os.listdir(classpath):
                folder = os.path.join(classpath, anomaly)
                files = [os.path.join(folder, x) for x in sorted(os.listdir(folder))]
                if self.train_val_split < 1.0:
                    cut = int(len(files) * self.train_val_split)
                    if self.split == DatasetSplit.TRAIN:
                        files = files[:cut]
                    elif self.split == DatasetSplit.VAL:
                        files = files[cut:]
                per_class[classname][anomaly] = files
                if self.split == DatasetSplit.TEST and anomaly != "good":
                    mfolder = os.path.join(maskpath, anomaly)
                    masks[classname][anomaly] = [os.path.join(mfolder, x) for x in sorted(os.listdir(mfolder))]
                else:
                    masks[classname]["good"] = None
        rows = []
        for classname in sorted(per_class.keys()):
            for anomaly in sorted(per_class[classname].keys()):
                for i, image_path in enumerate(per_class[classname][anomaly]):
                    with_mask = self.split == DatasetSplit.TEST and anomaly != "good"
                    rows.append([classname, anomaly, image_path, masks[classname][anomaly][i] if with_mask else None])
        return per_class, rows
