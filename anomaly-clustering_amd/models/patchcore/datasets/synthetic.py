"""MVTec-AD-shaped synthetic data (no network / no dataset on the build and GPU boxes).

Same item dict as ``patchcore.datasets.mvtec.MVTecDataset`` (keys image, mask, classname,
anomaly, is_anomaly, image_name, image_path), the real MVTec-AD test-set label histograms
(counted from the per-image records the reference ships under outputs/mvtec_ad/info/), and the
public train-set sizes.  Images: ``randn`` at ImageNet-normalised statistics from a per-category
generator, plus - so that clustering has a non-degenerate answer - a class-dependent localized
blob (class c > 0: a 32x32 patch at a class-specific location, amplitude 2.0).  "good" images and
all train images carry no blob.  (SURVEY.md 8d "Synthetic inputs".)
"""
from enum import Enum

import torch

# anomaly type -> count, in sorted (= dataset iteration) order, per MVTec-AD test split
TEST_HISTOGRAMS = {
    "bottle": [["broken_large", 20], ["broken_small", 22], ["contamination", 21], ["good", 20]],
    "cable": [["bent_wire", 13], ["cable_swap", 12], ["combined", 11], ["cut_inner_insulation", 14],
              ["cut_outer_insulation", 10], ["good", 58], ["missing_cable", 12], ["missing_wire", 10],
              ["poke_insulation", 10]],
    "capsule": [["crack", 23], ["faulty_imprint", 22], ["good", 23], ["poke", 21], ["scratch", 23], ["squeeze", 20]],
    "carpet": [["color", 19], ["cut", 17], ["good", 28], ["hole", 17], ["metal_contamination", 17], ["thread", 19]],
    "grid": [["bent", 12], ["broken", 12], ["glue", 11], ["good", 21], ["metal_contamination", 11], ["thread", 11]],
    "hazelnut": [["crack", 18], ["cut", 17], ["good", 40], ["hole", 18], ["print", 17]],
    "leather": [["color", 19], ["cut", 19], ["fold", 17], ["glue", 19], ["good", 32], ["poke", 18]],
    "metal_nut": [["bent", 25], ["color", 22], ["flip", 23], ["good", 22], ["scratch", 23]],
    "pill": [["color", 25], ["combined", 17], ["contamination", 21], ["crack", 26], ["faulty_imprint", 19],
             ["good", 26], ["pill_type", 9], ["scratch", 24]],
    "screw": [["good", 41], ["manipulated_front", 24], ["scratch_head", 24], ["scratch_neck", 25],
              ["thread_side", 23], ["thread_top", 23]],
    "tile": [["crack", 17], ["glue_strip", 18], ["good", 33], ["gray_stroke", 16], ["oil", 18], ["rough", 15]],
    "toothbrush": [["defective", 30], ["good", 12]],
    "transistor": [["bent_lead", 10], ["cut_lead", 10], ["damaged_case", 10], ["good", 60], ["misplaced", 10]],
    "wood": [["color", 8], ["combined", 11], ["good", 19], ["hole", 10], ["liquid", 10], ["scratch", 21]],
    "zipper": [["broken_teeth", 19], ["combined", 16], ["fabric_border", 17], ["fabric_interior", 16], ["good", 32],
               ["rough", 17], ["split_teeth", 18], ["squeezed_teeth", 16]],
}
TRAIN_SIZES = {"bottle": 209, "cable": 224, "capsule": 219, "carpet": 280, "grid": 264, "hazelnut": 391,
               "leather": 245, "metal_nut": 220, "pill": 267, "screw": 320, "tile": 230, "toothbrush": 60,
               "transistor": 213, "wood": 247, "zipper": 240}
_CLASSNAMES = list(TEST_HISTOGRAMS.keys())


class DatasetSplit(Enum):
    TRAIN = "train"
    VAL = "val"
    TEST = "test"


def category_labels(category, split=DatasetSplit.TEST, limit_per_class=None):
    """Per-image anomaly-type strings in dataset order."""
    if split != DatasetSplit.TEST:
        return ["good"] * TRAIN_SIZES[category]
    out = []
    for name, count in TEST_HISTOGRAMS[category]:
        out += [name] * (count if limit_per_class is None else min(count, limit_per_class))
    return out


def make_images(category, anomalies, imagesize=224, split=DatasetSplit.TEST, device="cpu"):
    """[N, 3, S, S] fp32 synthetic images for the given per-image anomaly names (deterministic)."""
    cat_idx = _CLASSNAMES.index(category) if category in _CLASSNAMES else len(_CLASSNAMES)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1234 + cat_idx + (100000 if split != DatasetSplit.TEST else 0))
    n = len(anomalies)
    images = torch.randn((n, 3, imagesize, imagesize), generator=gen, dtype=torch.float32)
    kinds = sorted(set(a for a in anomalies if a != "good"))
    cells = max(1, (imagesize - 32) // 32)
    for i, a in enumerate(anomalies):
        if a == "good":
            continue
        c = kinds.index(a) + 1
        y0 = 16 + 32 * ((5 * c) % cells)
        x0 = 16 + 32 * ((3 * c + 1) % cells)
        images[i, :, y0:y0 + 32, x0:x0 + 32] += 2.0
    return images.to(device)


class SyntheticMVTecDataset(torch.utils.data.Dataset):
    """Drop-in for MVTecDataset(source, classname, resize, imagesize, split) on synthetic pixels."""

    def __init__(self, source=None, classname="bottle", resize=256, imagesize=224, split=DatasetSplit.TRAIN,
                 train_val_split=1.0, limit_per_class=None, **kwargs):
        super().__init__()
        self.source = source
        self.split = split if isinstance(split, DatasetSplit) else DatasetSplit(getattr(split, "value", split))
        self.classname = classname
        self.imagesize = (3, imagesize, imagesize)
        self.anomalies = category_labels(classname, self.split, limit_per_class)
        self.images = make_images(classname, self.anomalies, imagesize, self.split)

    def __len__(self):
        return len(self.anomalies)

    def load_category(self, device, batch_size=64, only=None):
        """Same contract as MVTecDataset.load_category: (images on `device`, per-image records as collated by a
        batch-size-1 DataLoader) from one pass (`only` is accepted and ignored: the pixels already exist)."""
        info = []
        for idx in range(len(self)):
            item = self[idx]
            info.append({"classname": [item["classname"]], "anomaly": [item["anomaly"]],
                         "is_anomaly": torch.tensor([item["is_anomaly"]]), "image_name": [item["image_name"]],
                         "image_path": [item["image_path"]]})
        return self.images.to(device), info

    def __getitem__(self, idx):
        anomaly = self.anomalies[idx]
        name = "%s/%s/%s/%03d.png" % (self.classname, self.split.value, anomaly, idx)
        return {
            "image": self.images[idx],
            "mask": torch.zeros([1, *self.images.shape[-2:]]),
            "classname": self.classname,
            "anomaly": anomaly,
            "is_anomaly": int(anomaly != "good"),
            "image_name": name,
            "image_path": "synthetic://" + name,
        }
