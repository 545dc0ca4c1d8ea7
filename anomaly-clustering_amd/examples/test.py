"""Cluster + metrics consumer (drop-in for the reference's examples/test.py:134-345).

``calculate_metrics`` keeps the reference's signature and return value
``(NMI, ARI, F1, label, predict)``; it loads the ``(alpha, X)`` pickle written by main.py and the
``info_{category}.pickle`` records, drops "combined" images, runs Ward clustering with k = number
of labels (pairwise distances, Ward linkage and the NMI / ARI contingency matrix on the GPU; tree cut and
Hungarian relabelling of <= 9 clusters on the host) and the three scores.  The tau sweep writes the same CSV layout (encoding gbk; rows ``[mode]``, header, ``---``,
``TAU=..``, the 10 object and 5 texture categories, then the two size-weighted aggregates).

The reference reads ``../outputs/...`` relative to the working directory and has no CLI; here the
same literals are the defaults of a small argument parser.  The alpha heat-map dump of the reference
(test.py:158-174 + visualize() :71-105: one figure per anomaly type with the image, its ground-truth mask and the
28x28 alpha map blown up 8x by nearest neighbour and scaled to its maximum) is ``--visualize``: the same three
panels written with PIL instead of matplotlib, to ``<output_dir>/visualize/<run name>/<class>_<anomaly>.png``.
"""
import argparse
import csv
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_MODELS = os.path.join(os.path.dirname(_HERE), "models")
if _MODELS not in sys.path:
    sys.path.insert(0, _MODELS)

from patchcore.clustering import best_map, cluster_metrics  # noqa: E402,F401

os.environ["OMP_NUM_THREADS"] = '1'

_OBJECT = ["bottle", "cable", "capsule", "hazelnut", "metal_nut", "pill", "screw", "toothbrush", "transistor", "zipper"]
_TEXTURE = ["carpet", "grid", "leather", "tile", "wood"]
_CLASSNAMES = _OBJECT + _TEXTURE

OUTPUT_DIR = "../outputs"


def _safe_load(path):
    """torch.load(weights_only=True) with exactly the numpy globals the artefacts need."""
    import numpy
    allow = [numpy.ndarray, numpy.dtype, type(numpy.dtype("float32")), type(numpy.dtype("float64")),
             (numpy._core.multiarray._reconstruct, "numpy.core.multiarray._reconstruct"),
             numpy._core.multiarray._reconstruct]
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location='cpu', weights_only=True)


def artefact_path(category, pretrain_embed_dimension, target_embed_dimension, backbone_names, layers_to_extract_from,
                  train_ratio, tau, supervised, dataset, output_dir=None):
    root = OUTPUT_DIR if output_dir is None else output_dir
    return (root + "/" + dataset + "/" + backbone_names[0] + "/" + supervised + "/" +
            "_".join(layers_to_extract_from) + "_" + str(pretrain_embed_dimension) + "_" +
            str(target_embed_dimension) + "_" + str(float(tau)) + "_" + str(float(train_ratio)) +
            "/matrix_alpha_X_" + category + "_" + supervised + ".pickle")


def calculate_metrics(category, pretrain_embed_dimension, target_embed_dimension, backbone_names,
                      layers_to_extract_from, patchsize, train_ratio=1.0, tau=0.1, supervised="unsupervised",
                      dataset="mvtec_ad", output_dir=None):
    root = OUTPUT_DIR if output_dir is None else output_dir
    matrix_alpha, X = _safe_load(artefact_path(category, pretrain_embed_dimension, target_embed_dimension,
                                               backbone_names, layers_to_extract_from, train_ratio, tau, supervised,
                                               dataset, root))
    info = _safe_load(root + "/" + dataset + "/info/info_" + category + ".pickle")
    anomalies = [rec["anomaly"][0] for rec in info]
    NMI, ARI, F1, label, predict = cluster_metrics(X, anomalies)
    print(category)
    print(f'NMI: {NMI}')
    print(f'ARI: {ARI}')
    print(f'F1:{F1}\n')
    return NMI, ARI, F1, label, predict


def alpha_heatmap(alpha_row, scale=8):
    """Reference test.py:162-168: alpha [P] -> sqrt(P) x sqrt(P) map, nearest-neighbour x8, divided by its maximum;
    as the uint8 image ToPILImage makes of it (x255, truncated)."""
    a = np.asarray(alpha_row, dtype=np.float32).reshape(-1)
    side = int(round(len(a) ** 0.5))
    if side * side != len(a):
        raise ValueError("alpha row of %d weights is not a square patch grid" % len(a))
    m = a.max()
    grid = (a / m if m > 0 else a).reshape(side, side)
    return (np.kron(grid, np.ones((scale, scale), dtype=np.float32)) * 255.0).astype(np.uint8)


def render_heatmaps(category, matrix_alpha, info, out_dir, data_root=None):
    """One PNG per anomaly type (the first image of each, as the reference): [image | mask | alpha] panels of
    224 x 224.  Image and mask panels need the dataset on disk (data_root replaces nothing - info carries the
    paths); when a file is missing (synthetic data) the panel stays grey."""
    import PIL.Image
    os.makedirs(out_dir, exist_ok=True)
    alpha = matrix_alpha.squeeze(1) if matrix_alpha.dim() == 3 else matrix_alpha
    written, current = [], None
    for i, rec in enumerate(info):
        anomaly = rec["anomaly"][0]
        if anomaly == current:
            continue
        current = anomaly
        heat = alpha_heatmap(alpha[i].cpu().numpy())
        side = heat.shape[0]
        canvas = PIL.Image.new("RGB", (3 * side + 16, side), (127, 127, 127))
        path = rec["image_path"][0]

        def panel(p, mode):
            if not os.path.isfile(p):
                return None
            im = PIL.Image.open(p).convert(mode).resize((256, 256), PIL.Image.BILINEAR)      # Resize([256, 256])
            off = (256 - side) // 2
            return im.crop((off, off, off + side, off + side)).convert("RGB")                  # CenterCrop(224)
        img = panel(path, "RGB")
        if img is not None:
            canvas.paste(img, (0, 0))
        if anomaly != "good":
            mask = panel(path.replace("test", "ground_truth").replace(".png", "_mask.png"), "L")
            if mask is not None:
                canvas.paste(mask, (side + 8, 0))
        canvas.paste(PIL.Image.fromarray(heat, "L").convert("RGB"), (2 * side + 16, 0))
        fname = os.path.join(out_dir, rec["classname"][0] + "_" + anomaly + ".png")
        canvas.save(fname)
        written.append(fname)
    return written


def sweep(supervised_modes, tau_list, pretrain_embed_dimension, target_embed_dimension, backbone_names,
          layers_to_extract_from, patchsize, train_ratio, dataset="mvtec_ad", output_dir=None, variable="tau",
          objects=_OBJECT, textures=_TEXTURE):
    """The reference's __main__ loop (test.py:249-325): one CSV per mode."""
    root = OUTPUT_DIR if output_dir is None else output_dir
    written = []
    for supervised in supervised_modes:
        path = root + "/" + dataset + "/" + backbone_names[0] + "/" + supervised + "/"
        file_name = ("_".join(layers_to_extract_from) + "_" + str(pretrain_embed_dimension) + "_" +
                     str(target_embed_dimension) + "_" + variable + "_result.csv")
        with open(path + file_name, 'w', newline='', encoding='gbk') as csv_file:
            writer = csv.writer(csv_file)
            writer.writerow([supervised])
            writer.writerow(["Category", "NMI", "ARI", "F1"])
            for tau in tau_list:
                writer.writerow(["---"] * 4)
                writer.writerow(["TAU=" + str(tau)])
                for group_name, group in (("MVTec(object)", objects), ("MVTec(texture)", textures)):
                    acc, total = np.zeros(3), 0
                    rows = []
                    for category in group:
                        print("{:-^80}".format(category))
                        NMI, ARI, F1, label, predict = calculate_metrics(
                            category=category, pretrain_embed_dimension=pretrain_embed_dimension,
                            target_embed_dimension=target_embed_dimension, backbone_names=backbone_names,
                            layers_to_extract_from=layers_to_extract_from, patchsize=patchsize, tau=tau,
                            supervised=supervised, train_ratio=train_ratio, dataset=dataset, output_dir=root)
                        rows.append([category, NMI, ARI, F1])
                        acc += np.array([NMI, ARI, F1]) * len(label)
                        total += len(label)
                    for r in rows:
                        writer.writerow(r)
                    if group_name == "MVTec(object)":
                        pending = (group_name, acc / max(total, 1))
                    else:
                        # reference order: object rows, texture rows, then both aggregates
                        for name, vals in (pending, (group_name, acc / max(total, 1))):
                            print(name)
                            print(f'NMI: {vals[0]}')
                            print(f'ARI: {vals[1]}')
                            print(f'F1:{vals[2]}\n')
                            writer.writerow([name, vals[0], vals[1], vals[2]])
        written.append(path + file_name)
    return written


def main(argv=None):
    ap = argparse.ArgumentParser("Cluster the (alpha, X) artefacts of main.py and score them")
    ap.add_argument("--output_dir", default=OUTPUT_DIR)
    ap.add_argument("--dataset", default="mvtec_ad")
    ap.add_argument("--backbone_names", nargs="+", default=["dino_vitbase8"])
    ap.add_argument("--layers_to_extract_from", nargs="+", default=['blocks.10', 'blocks.11'])
    ap.add_argument("--pretrain_embed_dimension", type=int, default=2048)
    ap.add_argument("--target_embed_dimension", type=int, default=4096)
    ap.add_argument("--patchsize", type=int, default=3)
    ap.add_argument("--train_ratio", type=float, default=1)
    ap.add_argument("--tau_list", nargs="+", type=float,
                    default=[0, 0.2, 0.4, 0.6, 0.8, 1, 1.5, 2, 2.5, 3, 4, 8, 10, 12, 14, 18, 20])
    ap.add_argument("--supervised", nargs="+", default=["unsupervised", "supervised"])
    ap.add_argument("--objects", nargs="*", default=_OBJECT)
    ap.add_argument("--textures", nargs="*", default=_TEXTURE)
    ap.add_argument("--visualize", action="store_true", help="also dump the alpha heat-maps (reference test.py:158-174)")
    a = ap.parse_args(argv)
    taus = [int(t) if float(t).is_integer() else t for t in a.tau_list]     # "TAU=2", not "TAU=2.0", as upstream
    if a.visualize:
        for supervised in a.supervised:
            for tau in taus:
                for category in list(a.objects) + list(a.textures):
                    alpha, _ = _safe_load(artefact_path(category, a.pretrain_embed_dimension, a.target_embed_dimension,
                                                        a.backbone_names, a.layers_to_extract_from, a.train_ratio, tau,
                                                        supervised, a.dataset, a.output_dir))
                    info = _safe_load(a.output_dir + "/" + a.dataset + "/info/info_" + category + ".pickle")
                    name = (a.dataset + "/" + a.backbone_names[0] + "_" + str(a.pretrain_embed_dimension) + "_" +
                            str(a.target_embed_dimension) + "_" + "_".join(a.layers_to_extract_from) + "_" +
                            str(float(tau)) + "_" + supervised)
                    render_heatmaps(category, alpha, info, os.path.join(a.output_dir, "visualize", name))
    return sweep(a.supervised, taus, a.pretrain_embed_dimension, a.target_embed_dimension, a.backbone_names,
                 a.layers_to_extract_from, a.patchsize, a.train_ratio, a.dataset, a.output_dir,
                 objects=a.objects, textures=a.textures)


if __name__ == "__main__":
    main()
