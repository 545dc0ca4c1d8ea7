#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference, read-only); nothing
here runs on the GPU box.  It imports the reference's own modules
(patchcore.vision_transformer / common / patchcore / utils) with three empty
import-time stubs for packages that are absent here and never touched on the
path (faiss, timm, torchvision - SURVEY.md 8c), feeds them RNG-free inputs
from tests/_synth.py, and stores what they return.  Stored: outputs only
(plus per-tensor checksums of the seed-0 weights); no reference source.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
    python tests/golden/make_golden.py --kat-all  # kat_all_pdist.npz: all 26 committed (alpha, X) artefacts (data files only)
"""
import csv
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Anomaly-Clustering"
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _synth import hash_tensor, gaussish_tensor, clustered_embeddings, dino_like_statistics  # noqa: E402


def import_reference():
    faiss = types.ModuleType("faiss")
    faiss.omp_set_num_threads = lambda n: None
    sys.modules["faiss"] = faiss
    sys.modules["timm"] = types.ModuleType("timm")
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    sys.path.insert(0, os.path.join(REF, "models"))
    import patchcore.vision_transformer as vits
    import patchcore.common as common
    import patchcore.patchcore as pc
    import patchcore.utils as utils
    import patchcore.sampler as sampler
    return vits, common, pc, utils, sampler


def checksums(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), (t * t).sum().item(), t[0].item(), t[-1].item()])


PATCH_ROWS = [0, 1, 27, 28, 29, 55, 391, 392, 406, 419, 420, 756, 757, 782, 783]   # corners, edges, centre
TOKENS = [0, 1, 2, 28, 29, 392, 783, 784]


def a15_fixture():
    """A small RGB image (RNG-free content: smooth gradients + blocks + hash noise) saved as PNG, and what the
    reference's transform chain returns for it.  torchvision is absent here; for PIL inputs its Resize(256) /
    CenterCrop(224) are Pillow's Image.resize(BILINEAR) / crop (sizes and offsets as in torchvision's
    functional.py), ToTensor / Normalize are torch ops - the chain below IS those calls."""
    import PIL.Image
    h, w = 300, 340
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 23.0) * np.cos(yy / 31.0), (xx * 255.0 / w), ((xx // 40 + yy // 25) % 2) * 180.0 + 30], -1)
    noise = hash_tensor((h, w, 3), salt=77, scale=40.0, dtype=np.float64)
    img = np.clip(base + noise, 0, 255).astype(np.uint8)
    PIL.Image.fromarray(img).save(os.path.join(HERE, "a15_image.png"), optimize=True)
    pil = PIL.Image.open(os.path.join(HERE, "a15_image.png")).convert("RGB")
    nw, nh = (256, int(256 * h / w)) if w <= h else (int(256 * w / h), 256)
    r = pil.resize((nw, nh), PIL.Image.BILINEAR)
    top, left = int(round((nh - 224) / 2.0)), int(round((nw - 224) / 2.0))
    r = r.crop((left, top, left + 224, top + 224))
    u8 = np.asarray(r)
    t = torch.from_numpy(np.ascontiguousarray(u8)).permute(2, 0, 1).to(torch.float32).div(255)
    t = (t - torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    np.savez_compressed(os.path.join(HERE, "a15_expected.npz"), shape=np.array(img.shape),
                        cropped_u8=np.transpose(u8, (2, 0, 1)), check=checksums(t))
    print("a15 fixture", img.shape, (nh, nw), (top, left), checksums(t))


def main(only_kat=False, only_dino=False):
    torch.set_num_threads(8)
    vits, common, pc, utils, sampler = import_reference()
    dev = torch.device("cpu")
    if only_dino:
        return dino_stats_golden(vits, common, dev)
    if not only_kat:
        compute_goldens(vits, common, pc, utils, sampler, dev)
    known_answer_rows()


def dino_stats_golden(vits, common, dev):
    """Only G1b (see compute_goldens): written separately so that the other fixtures need not be regenerated."""
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=11, scale=1.7))
    torch.manual_seed(0)
    model_d = vits.vit_base(patch_size=8, num_classes=0)
    for p in model_d.parameters():
        p.requires_grad = False
    model_d.eval()
    sd_d = model_d.state_dict()
    dino_like_statistics(sd_d)
    agg_d = common.NetworkFeatureAggregator(model_d, ["blocks.10", "blocks.11"], dev)
    feats_d = agg_d(images)
    d10, d11 = feats_d["blocks.10"].clone(), feats_d["blocks.11"].clone()
    names_d = sorted(sd_d.keys())
    np.savez(os.path.join(HERE, "vit_forward_dino_stats.npz"),
             tokens=np.array(TOKENS), f10=d10[:, TOKENS].numpy(), f11=d11[:, TOKENS].numpy(),
             f10_check=checksums(d10), f11_check=checksums(d11),
             weight_checks=np.stack([checksums(sd_d[k]) for k in names_d]),
             absmax=np.array([d10.abs().max().item(), d11.abs().max().item()]))
    print("G1b done", checksums(d10), checksums(d11), d10.abs().max().item(), d11.abs().max().item())


def compute_goldens(vits, common, pc, utils, sampler, dev):

    # ---------------- G1: ViT-B/8, seed-0 init, two hash images -----------------
    torch.manual_seed(0)
    model = vits.vit_base(patch_size=8, num_classes=0)
    for p in model.parameters():
        p.requires_grad = False
    model.eval()
    sd = model.state_dict()
    names = sorted(sd.keys())
    np.savez(os.path.join(HERE, "vit_init_seed0.npz"),
             names=np.array(names), checks=np.stack([checksums(sd[k]) for k in names]),
             shapes=np.array([str(tuple(sd[k].shape)) for k in names]))
    images = torch.from_numpy(gaussish_tensor((2, 3, 224, 224), salt=11, scale=1.7))
    agg = common.NetworkFeatureAggregator(model, ["blocks.10", "blocks.11"], dev)
    feats = agg(images)
    f10, f11 = feats["blocks.10"].clone(), feats["blocks.11"].clone()
    assert f10.shape == (2, 785, 768)
    fdims = agg.feature_dimensions((3, 224, 224))
    np.savez(os.path.join(HERE, "vit_forward.npz"),
             tokens=np.array(TOKENS), f10=f10[:, TOKENS].numpy(), f11=f11[:, TOKENS].numpy(),
             f10_check=checksums(f10), f11_check=checksums(f11), feature_dimensions=np.array(fdims))
    print("G1 done", checksums(f10), checksums(f11))

    # ---------------- G1b: the same network with DINO-like statistics ------------
    dino_stats_golden(vits, common, dev)
    agg = common.NetworkFeatureAggregator(model, ["blocks.10", "blocks.11"], dev)     # hooks back on the plain model

    # ---------------- G2: aggregation through the reference's _embed -------------
    core = pc.AnomalyClusteringCore(dev)
    core.load(backbone=model, layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096,
              patchsize=3, featuresampler=sampler.IdentitySampler(), anomaly_scorer_num_nn=5,
              nn_method=common.FaissNN(False, 4))
    Z = torch.from_numpy(np.stack(core._embed(images, "unsupervised")))       # list of rows -> [2*784, 4096]
    assert Z.shape == (2 * 784, 4096)
    rows = PATCH_ROWS + [784 + r for r in PATCH_ROWS]
    np.savez(os.path.join(HERE, "aggregate_vit.npz"), rows=np.array(rows), Z_rows=Z[rows].numpy(),
             Z_check=checksums(Z), Z_colsum=Z.double().sum(0).numpy(), Z_rowsum=Z.double().sum(1).numpy())
    print("G2 done", checksums(Z))

    # G2-variants: aggregation alone on hash feature maps (no ViT), other dims / patch sizes.
    class Taps(torch.nn.Module):
        """Backbone stand-in that returns fixed feature maps through the same hook API."""
        def __init__(self, fa, fb):
            super().__init__()
            self.layer2 = torch.nn.Identity()
            self.layer3 = torch.nn.Identity()
            self.fa, self.fb = fa, fb

        def forward(self, x):
            self.layer2(self.fa[: x.shape[0]])
            self.layer3(self.fb[: x.shape[0]])
            return x

    def run_variant(tag, fa, fb, patchsize, pre, tgt):
        bb = Taps(fa, fb)
        c = pc.AnomalyClusteringCore(dev)
        c.load(backbone=bb, layers_to_extract_from=["layer2", "layer3"], device=dev,
               input_shape=(3, 224, 224), pretrain_embed_dimension=pre, target_embed_dimension=tgt,
               patchsize=patchsize, featuresampler=sampler.IdentitySampler(),
               nn_method=common.FaissNN(False, 4))
        z = torch.from_numpy(np.stack(c._embed(torch.zeros(fa.shape[0], 3, 224, 224), "x")))
        np.savez(os.path.join(HERE, "aggregate_%s.npz" % tag), Z=z.numpy(),
                 patchsize=patchsize, pre=pre, tgt=tgt)
        print("variant", tag, tuple(z.shape), checksums(z))

    # (a) token maps, small: T=1+6*6, C=64; 3x3; 192->128 per layer, 256->160 (real final pool)
    ta = torch.from_numpy(gaussish_tensor((2, 37, 64), salt=21, scale=2.0)) + 0.3
    tb = torch.from_numpy(gaussish_tensor((2, 37, 64), salt=22, scale=1.0)) - 0.1
    run_variant("tok_small", ta, tb, 3, 128, 160)
    # (b) same maps, patchsize 5, 200 -> 400 (up-"pooling" windows of width 1/2)
    run_variant("tok_p5", ta, tb, 5, 200, 400)
    # (c) CNN branch: [2,32,12,12] + [2,48,6,6] -> bilinear-aligned; 3x3; 96 -> 96
    ca = torch.from_numpy(gaussish_tensor((2, 32, 12, 12), salt=23, scale=2.0))
    cb = torch.from_numpy(gaussish_tensor((2, 48, 6, 6), salt=24, scale=1.5))
    run_variant("cnn_small", ca, cb, 3, 96, 96)
    # (d) WRN50-shaped CNN branch (config 1): [1,512,28,28] + [1,1024,14,14], 1024 -> 1024; rows subset
    wa = torch.from_numpy(gaussish_tensor((1, 512, 28, 28), salt=25, scale=2.0))
    wb = torch.from_numpy(gaussish_tensor((1, 1024, 14, 14), salt=26, scale=1.0))
    bb = Taps(wa, wb)
    c = pc.AnomalyClusteringCore(dev)
    c.load(backbone=bb, layers_to_extract_from=["layer2", "layer3"], device=dev, input_shape=(3, 224, 224),
           pretrain_embed_dimension=1024, target_embed_dimension=1024, patchsize=3,
           featuresampler=sampler.IdentitySampler(), nn_method=common.FaissNN(False, 4))
    zw = torch.from_numpy(np.stack(c._embed(torch.zeros(1, 3, 224, 224), "x")))
    np.savez(os.path.join(HERE, "aggregate_wrn.npz"), rows=np.array(PATCH_ROWS), Z_rows=zw[PATCH_ROWS].numpy(),
             Z_check=checksums(zw), Z_rowsum=zw.double().sum(1).numpy())
    print("variant wrn", checksums(zw))

    # ---------------- G3: alpha (small, all taus, both modes, tie case) ----------
    taus = [0, 0.2, 1, 2, 20]
    Zs, _ = clustered_embeddings(6, 49, 128, 3, salt=31)
    Zt, _ = clustered_embeddings(4, 49, 128, 1, salt=32)
    Zs_t, Zt_t = torch.from_numpy(Zs), torch.from_numpy(Zt)
    out = {"taus": np.array(taus, dtype=np.float64)}
    out["w_unsup"] = np.stack([utils.Weight_Distance_Unsupervised(Zs_t, i, dev).numpy() for i in range(6)])
    out["w_sup"] = np.stack([utils.Weight_Distance_Supervised(Zs_t, Zt_t, i, dev).numpy() for i in range(6)])
    for t in taus:
        out["alpha_unsup_%g" % t] = utils.Matrix_Alpha_Unsupervised(t, 1, Zs_t, dev).numpy()
        out["alpha_sup_%g" % t] = utils.Matrix_Alpha_Supervised(t, 1, Zs_t, Zt_t, dev).numpy()
    # tie case for tau=0: image 0 and image 1 identical -> supervised against a bank that
    # contains two copies of the same patch rows gives exact ties in w
    Ztie = Zs.copy()
    Ztie[0, 5] = Ztie[0, 9]            # two identical patches in the query -> identical w -> tie if maximal
    Ztie[0, 5] += 40.0                 # make them the far-away (maximal) ones
    Ztie[0, 9] = Ztie[0, 5]
    Ztie_t = torch.from_numpy(Ztie)
    out["alpha_tie_unsup_0"] = utils.Matrix_Alpha_Unsupervised(0, 1, Ztie_t, dev).numpy()
    out["alpha_tie_sup_0"] = utils.Matrix_Alpha_Supervised(0, 1, Ztie_t, Zt_t, dev).numpy()
    assert (out["alpha_tie_unsup_0"][0] == 0.5).sum() == 2, out["alpha_tie_unsup_0"][0]
    # X for the tau=1 unsupervised alpha (main.py:294-296)
    a = torch.from_numpy(out["alpha_unsup_1"]).unsqueeze(1).float()
    out["X_unsup_1"] = np.array(torch.bmm(a, Zs_t).squeeze(1))
    np.savez(os.path.join(HERE, "alpha_small.npz"), **out)
    print("G3 small done")

    # full-size P=784, D=4096, N=3 queries, M=2 bank
    Zf, _ = clustered_embeddings(3, 784, 4096, 3, salt=41)
    Zb, _ = clustered_embeddings(2, 784, 4096, 1, salt=42)
    Zf_t, Zb_t = torch.from_numpy(Zf), torch.from_numpy(Zb)
    full = {
        "w_unsup": np.stack([utils.Weight_Distance_Unsupervised(Zf_t, i, dev).numpy() for i in range(3)]),
        "w_sup": np.stack([utils.Weight_Distance_Supervised(Zf_t, Zb_t, i, dev).numpy() for i in range(3)]),
        "alpha_unsup_1": utils.Matrix_Alpha_Unsupervised(1, 1, Zf_t, dev).numpy(),
        "alpha_sup_1": utils.Matrix_Alpha_Supervised(1, 1, Zf_t, Zb_t, dev).numpy(),
        "alpha_unsup_0.2": utils.Matrix_Alpha_Unsupervised(0.2, 1, Zf_t, dev).numpy(),
    }
    a = torch.from_numpy(full["alpha_unsup_1"]).unsqueeze(1).float()
    full["X_unsup_1"] = np.array(torch.bmm(a, Zf_t).squeeze(1))
    np.savez(os.path.join(HERE, "alpha_full.npz"), **full)
    print("G3 full done")



def known_answer_rows():
    # ---------------- G5: known-answer rows from the reference's committed artefacts ----
    import numpy
    # the pickles name numpy globals by their numpy-1.x paths; allow-list exactly those three
    # (never weights_only=False: the pickles are untrusted content)
    allow = [numpy.ndarray, numpy.dtype, type(numpy.dtype("float32")),
             (numpy._core.multiarray._reconstruct, "numpy.core.multiarray._reconstruct")]
    outdir = os.path.join(REF, "outputs/mvtec_ad")

    def csv_rows(mode):
        rows, tau = {}, None
        with open(os.path.join(outdir, "dino_vitbase8", mode, "blocks.10_blocks.11_2048_4096_tau_result.csv"),
                  encoding="gbk") as f:
            for r in csv.reader(f):
                if r and r[0].startswith("TAU="):
                    tau = r[0][4:]
                elif tau is not None and len(r) == 4 and r[0] not in ("---", "Category"):
                    rows[(tau, r[0])] = [float(v) for v in r[1:]]
        return rows

    for mode, cats in (("unsupervised", ["toothbrush", "grid"]), ("supervised", ["toothbrush"])):
        rows = csv_rows(mode)
        for cat in cats:
            with torch.serialization.safe_globals(allow):
                alpha, X = torch.load(os.path.join(outdir, "dino_vitbase8", mode,
                                                   "blocks.10_blocks.11_2048_4096_2.0_1.0",
                                                   "matrix_alpha_X_%s_%s.pickle" % (cat, mode)),
                                      map_location="cpu", weights_only=True)
            info = torch.load(os.path.join(outdir, "info", "info_%s.pickle" % cat), map_location="cpu",
                              weights_only=True)
            anomalies = np.array([d["anomaly"][0] for d in info])
            np.savez_compressed(os.path.join(HERE, "kat_%s_%s.npz" % (cat, mode)), X=np.asarray(X, dtype=np.float32),
                                anomaly=anomalies, expected=np.array(rows[("2", cat)]),
                                alpha_rowsum=alpha.squeeze(1).double().sum(1).numpy())
            print("KAT", cat, mode, X.shape, rows[("2", cat)])


def known_answer_rows_all():
    """G5, complete (round 6): EVERY (alpha, X) artefact the reference commits - 13 categories x {unsupervised, supervised}
    at tau = 2 (pill and screw have none) - as what the cluster stage starts from: the condensed float64 Euclidean
    distances of the kept images (scipy's pdist on the float64 X, "combined" images dropped: examples/test.py:176-195),
    their anomaly-type strings and the reference's CSV row.  One file, derived arrays only (the three kat_*.npz above
    keep the full X for the pdist kernel)."""
    import numpy
    from scipy.spatial.distance import pdist
    allow = [numpy.ndarray, numpy.dtype, type(numpy.dtype("float32")),
             (numpy._core.multiarray._reconstruct, "numpy.core.multiarray._reconstruct")]
    outdir = os.path.join(REF, "outputs/mvtec_ad")
    out, n_rows = {}, 0
    for mode in ("unsupervised", "supervised"):
        rows, tau = {}, None
        with open(os.path.join(outdir, "dino_vitbase8", mode, "blocks.10_blocks.11_2048_4096_tau_result.csv"), encoding="gbk") as f:
            for r in csv.reader(f):
                if r and r[0].startswith("TAU="):
                    tau = r[0][4:]
                elif tau is not None and len(r) == 4 and r[0] not in ("---", "Category"):
                    rows[(tau, r[0])] = [float(v) for v in r[1:]]
        folder = os.path.join(outdir, "dino_vitbase8", mode, "blocks.10_blocks.11_2048_4096_2.0_1.0")
        for fn in sorted(os.listdir(folder)):
            if not (fn.startswith("matrix_alpha_X_") and fn.endswith("_%s.pickle" % mode)):
                continue
            cat = fn[len("matrix_alpha_X_"): -len("_%s.pickle" % mode)]
            with torch.serialization.safe_globals(allow):
                _, X = torch.load(os.path.join(folder, fn), map_location="cpu", weights_only=True)
            info = torch.load(os.path.join(outdir, "info", "info_%s.pickle" % cat), map_location="cpu", weights_only=True)
            names = [d["anomaly"][0] for d in info]
            keep = [i for i, a in enumerate(names) if a != "combined"]
            X64 = np.asarray(X)[keep].astype(np.float64)
            out["pdist_%s_%s" % (cat, mode)] = pdist(X64, metric="euclidean")
            out["names_%s" % cat] = np.array([names[i] for i in keep])
            out["expected_%s_%s" % (cat, mode)] = np.array(rows[("2", cat)])
            n_rows += 1
            print("KAT", cat, mode, X64.shape, rows[("2", cat)])
    out["rows"] = np.array(sorted(k[len("pdist_"):] for k in out if k.startswith("pdist_")))
    np.savez_compressed(os.path.join(HERE, "kat_all_pdist.npz"), **out)
    print("known-answer rows:", n_rows)


if __name__ == "__main__":
    if "--kat-all" in sys.argv:
        known_answer_rows_all()
    elif "--a15" in sys.argv:
        a15_fixture()
    else:
        main(only_kat="--kat" in sys.argv, only_dino="--dino" in sys.argv)
