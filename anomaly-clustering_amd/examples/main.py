"""alpha + X producer (drop-in for the reference's examples/main.py:183-366).

Same flags and defaults as the reference (--path --backbone_names --layers_to_extract_from
--pretrain_embed_dimension --target_embed_dimension --output_dir --patchsize --tau --train_ratio
--supervised --dataset), the same artefact: ``torch.save((alpha FloatTensor[N,1,P], X float32
ndarray[N,D]))`` under
``{output_dir}/{dataset}/{backbone}/{mode}/{layers}_{pre}_{tgt}_{float(tau)}_{float(ratio)}/
matrix_alpha_X_{category}_{mode}.pickle``, plus ``{output_dir}/{dataset}/info/info_{category}.pickle``
(the reference's writer is commented out at main.py:263 although test.py:156 needs the file).

Conscious differences:
  * --supervised selects the mode (reference overrides it with a hard-coded loop over both,
    main.py:348); ``--supervised both`` reproduces the reference's behaviour.
  * ``--dataset synthetic`` uses MVTec-AD-shaped synthetic images (only then: a mistyped --path fails).
  * images are embedded in batches and stay on the device; the normal-image bank of the
    supervised mode is a seeded permutation (reference: unseeded shuffle, main.py:274).
  * launched under torchrun, the images of each category are sharded over the ranks (RCCL).
  * the categories run as ONE job (``--category_schedule pipeline``, pipeline.run_categories): the rank's images of all
    categories share full ViT batches, every ring of shards is in flight before the alpha work starts.
    ``sequential`` is the reference's category-by-category loop (main.py:346-366), ``per_gpu`` gives whole categories
    to GPUs.  The artefacts are the same files with the same contents.
Run with ``PYTHONPATH=../models python main.py ...`` exactly like the reference.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
from tqdm import tqdm

_HERE = os.path.dirname(os.path.abspath(__file__))
_MODELS = os.path.join(os.path.dirname(_HERE), "models")
if _MODELS not in sys.path:
    sys.path.insert(0, _MODELS)

import patchcore  # noqa: E402
import patchcore.backbones  # noqa: E402
import patchcore.common  # noqa: E402
import patchcore.sampler  # noqa: E402
import patchcore.datasets.mvtec as mvtec  # noqa: E402
import patchcore.datasets.synthetic as synthetic  # noqa: E402
from patchcore.patchcore import AnomalyClusteringCore  # noqa: E402
from patchcore.pipeline import HipOps, category_rot, run_categories, run_category, select_bank, shard_bounds  # noqa: E402

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (the host driver's only mode): before the first GPU call
device = torch.device("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.is_available() else "cpu")

_CLASSNAMES = [
    "bottle", "cable", "capsule", "carpet", "grid", "hazelnut", "leather", "metal_nut", "pill", "screw",
    "tile", "toothbrush", "transistor", "wood", "zipper",
]


def artefact_dir(save_path, layers_to_extract_from, pretrain_embed_dimension, target_embed_dimension, tau, train_ratio):
    """Path scheme of the reference (main.py:298-301)."""
    return (save_path + "/" + "_".join(layers_to_extract_from) + "_" + str(pretrain_embed_dimension) + "_" +
            str(target_embed_dimension) + "_" + str(float(tau)) + "_" + str(float(train_ratio)))


def _dataset(path, category, split, dataset):
    if dataset != "synthetic":          # a missing --path raises (mvtec.py), as the reference would
        return mvtec.MVTecDataset(source=path, split=split, classname=category, resize=256, imagesize=224)
    return synthetic.SyntheticMVTecDataset(source=path, split=synthetic.DatasetSplit(split.value),
                                           classname=category, resize=256, imagesize=224)


def _build_core(backbone_names, layers_to_extract_from, pretrain_embed_dimension, target_embed_dimension, patchsize,
                backbone=None):
    """The model half of make_category_data (main.py:198-252): backbone -> AnomalyClusteringCore.load."""
    faiss_on_gpu = True
    faiss_num_workers = 4
    input_shape = (3, 224, 224)
    anomaly_scorer_num_nn = 5
    sampler = patchcore.sampler.IdentitySampler()
    backbone_seed = None
    backbone_name = backbone_names[0]
    if len(backbone_names) > 1:
        coll = [[] for _ in range(len(backbone_names))]
        for layer in layers_to_extract_from:
            idx = int(layer.split(".")[0])
            coll[idx].append(".".join(layer.split(".")[1:]))
    else:
        coll = [layers_to_extract_from]
    layers_to_extract_from = coll[0]
    if ".seed-" in backbone_name:
        backbone_name, backbone_seed = backbone_name.split(".seed-")[0], int(backbone_name.split("-")[-1])
    if backbone is None:
        backbone = patchcore.backbones.load(backbone_name)
    backbone.name, backbone.seed = backbone_name, backbone_seed
    nn_method = patchcore.common.FaissNN(faiss_on_gpu, faiss_num_workers)
    core = AnomalyClusteringCore(device)
    core.load(backbone=backbone, layers_to_extract_from=layers_to_extract_from, device=device,
              input_shape=input_shape, pretrain_embed_dimension=pretrain_embed_dimension,
              target_embed_dimension=target_embed_dimension, patchsize=patchsize, featuresampler=sampler,
              anomaly_scorer_num_nn=anomaly_scorer_num_nn, nn_method=nn_method)
    return core, layers_to_extract_from


def _ranks():
    multi = dist.is_available() and dist.is_initialized()
    return (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)


def _load_test(path, category, dataset, cache, whole=False, rot=0):
    """ONE pass over the test split: decode on the host, resize / crop / normalise on the GPU, and collect the per-image
    records (what test.py reads as `info`: the item dicts minus pixels, collated with batch size 1) on the way.
    The reference walks the test set twice with batch size 1 (main.py:254-262, :266).  `cache` (a dict the caller keeps
    per category): the loaded test set and its embeddings Z are the same for every mode, so `--supervised both`
    decodes and embeds the test images once.  Launched under torchrun, a rank decodes only the files of its own
    image shard (records of all); whole=True: all of them (a rank that may be given the WHOLE category).  rot: the
    shard is that of virtual rank (rank + rot) % world (pipeline.run_categories rotates the shards per category)."""
    rank, world = _ranks()
    rank = (rank + rot) % world
    if "images" not in cache:
        test_dataset = _dataset(path, category, mvtec.DatasetSplit.TEST, dataset)
        cache["n_test"] = len(test_dataset)
        cache["images"], cache["info"] = test_dataset.load_category(
            device, only=shard_bounds(len(test_dataset), world)[rank] if (world > 1 and not whole) else None)
    return cache["images"], cache["info"]


def _load_bank(path, category, dataset, n_test, train_ratio, whole=False, rot=0):
    """The train images the supervised mode may select (main.py:269-281); under torchrun only this rank's slice of
    the seeded selection is decoded (rot: see _load_test)."""
    rank, world = _ranks()
    rank = (rank + rot) % world
    train_ds = _dataset(path, category, mvtec.DatasetSplit.TRAIN, dataset)
    only = None
    if world > 1 and not whole:                     # the bank images this rank embeds: its slice of the seeded selection
        sel = select_bank(len(train_ds), n_test, train_ratio, 0)
        blo, bhi = shard_bounds(len(sel), world)[rank]
        only = set(int(i) for i in sel[blo:bhi])
    train_images, _ = train_ds.load_category(device, only=only)
    return train_images


def _save(out, info, category, supervised, tau, save_path, layers_to_extract_from, pretrain_embed_dimension,
          target_embed_dimension, train_ratio, info_dir):
    """torch.save((alpha, X)) per tau under the reference's path scheme (main.py:298-309) + the info records."""
    taus = list(tau) if isinstance(tau, (list, tuple)) else [tau]
    rank0 = _ranks()[0] == 0
    data_matrix = None
    for tv in taus:
        res = out["by_tau"][float(tv)]
        this = (res["alpha"].cpu(), res["X"])
        data_matrix = data_matrix or this
        if rank0:
            folder = artefact_dir(save_path, layers_to_extract_from, pretrain_embed_dimension, target_embed_dimension,
                                  tv, train_ratio)
            os.makedirs(folder, exist_ok=True)
            torch.save(this, folder + "/matrix_alpha_X_" + category + "_" + supervised + ".pickle")
    if rank0 and info_dir is not None:
        os.makedirs(info_dir, exist_ok=True)
        torch.save(info, os.path.join(info_dir, "info_" + category + ".pickle"))
    return data_matrix


def make_category_data(path, category, pretrain_embed_dimension, target_embed_dimension, backbone_names,
                       layers_to_extract_from, patchsize, save_path, train_ratio=1.0, tau=1, supervised="unsupervised",
                       dataset="mvtec_ad", info_dir=None, backbone=None, batch_size=32, cache=None):
    """One category, as the reference runs it (main.py:183-311)."""
    print("{:-^80}".format(category + ' start ' + supervised))
    core, layers = _build_core(backbone_names, layers_to_extract_from, pretrain_embed_dimension, target_embed_dimension,
                               patchsize, backbone)
    cache = {} if cache is None else cache
    images, info = _load_test(path, category, dataset, cache)
    train_images = _load_bank(path, category, dataset, cache["n_test"], train_ratio) if supervised == "supervised" else None
    out = run_category(HipOps(core, batch_size=batch_size), images, supervised, tau=tau, k=1,
                       train_images=train_images, train_ratio=train_ratio, want_pdist=False, Z_loc=cache.get("Z_loc"))
    cache["Z_loc"] = out["Z_loc"]
    # `tau` may be a list (a sweep): embeddings and patch distances were computed once; one artefact per tau
    data_matrix = _save(out, info, category, supervised, tau, save_path, layers, pretrain_embed_dimension,
                        target_embed_dimension, train_ratio, info_dir)
    print("{:-^80}".format(category + ' end'))
    return data_matrix


def make_categories_data(path, categories, pretrain_embed_dimension, target_embed_dimension, backbone_names,
                         layers_to_extract_from, patchsize, save_paths, train_ratio=1.0, tau=1, modes=("unsupervised",),
                         dataset="mvtec_ad", info_dir=None, backbone=None, batch_size=128, schedule="pipeline"):
    """All categories as ONE job per mode (pipeline.run_categories) - the same artefacts as make_category_data per
    category (the reference's loop, main.py:346-366).  save_paths: {mode: folder}.  -> {(mode, category): (alpha, X)}."""
    core, layers = _build_core(backbone_names, layers_to_extract_from, pretrain_embed_dimension, target_embed_dimension,
                               patchsize, backbone)
    ops = HipOps(core, batch_size=batch_size)
    whole = schedule == "per_gpu"                       # a rank may be handed any WHOLE category
    caches = [{} for _ in categories]
    items = []
    world = _ranks()[1]
    rots = [category_rot(c, world) if schedule == "pipeline" else 0 for c in range(len(categories))]
    for category, cache, rot in zip(categories, caches, rots):
        print("{:-^80}".format(category + ' load'))
        images, _ = _load_test(path, category, dataset, cache, whole=whole, rot=rot)
        items.append({"images": images, "train_images": None})
    results = {}
    Z_locs = None
    for supervised in modes:
        print("{:-^80}".format("%d categories start %s" % (len(categories), supervised)))
        if supervised == "supervised":
            for category, cache, it, rot in zip(categories, caches, items, rots):
                it["train_images"] = _load_bank(path, category, dataset, cache["n_test"], train_ratio, whole=whole, rot=rot)
        outs = run_categories(ops, items, supervised, tau=tau, k=1, train_ratio=train_ratio, want_pdist=False,
                              schedule=schedule, Z_locs=Z_locs if schedule != "per_gpu" else None)
        if schedule != "per_gpu":
            Z_locs = [o["Z_loc"] for o in outs]       # Z does not depend on the mode: the next mode embeds only its bank
        for category, cache, o in zip(categories, caches, outs):
            results[(supervised, category)] = _save(o, cache["info"], category, supervised, tau, save_paths[supervised],
                                                    layers, pretrain_embed_dimension, target_embed_dimension, train_ratio,
                                                    info_dir)
        for it in items:
            it["train_images"] = None
        print("{:-^80}".format("%d categories end" % len(categories)))
    return results


def build_parser():
    parser = argparse.ArgumentParser('Calculating Matrix on MVTec AD')
    parser.add_argument('--path', default='data/mvtec_ad', type=str, help="Path to the dataset.")
    parser.add_argument('--backbone_names', nargs='+', default=["dino_vitbase8"], help='Architecture.')
    parser.add_argument('--layers_to_extract_from', nargs='+', default=["blocks.10", "blocks.11"])
    parser.add_argument('--pretrain_embed_dimension', default=2048, type=int, help='Pretrained Embedding Dimension')
    parser.add_argument('--target_embed_dimension', default=4096, type=int, help='Target Embedding Dimension')
    parser.add_argument('--output_dir', default="outputs", help='Path where to save segmentations')
    parser.add_argument("--patchsize", type=int, default=3, help="Patch Size.")
    parser.add_argument("--tau", type=float, default=1, help="Tau.")
    parser.add_argument("--tau_list", nargs="*", type=float, default=None,
                        help="Sweep: several tau values from ONE embedding + patch-distance pass per category "
                             "(the reference's test.py reads 0 0.2 0.4 0.6 0.8 1 1.5 2 2.5 3 4 8 10 12 14 18 20).")
    parser.add_argument("--train_ratio", type=float, default=1, help="The ratio of train data.")
    parser.add_argument('--supervised', default='unsupervised', type=str,
                        help="unsupervised | supervised | average | both")
    parser.add_argument('--dataset', default='mvtec_ad', type=str, help="Dataset to use (mvtec_ad | synthetic).")
    parser.add_argument('--categories', nargs='*', default=None, help="Subset of categories (default: all 15).")
    parser.add_argument('--batch_size', type=int, default=256,
                        help="Largest ViT launch, in images (longer streams are cut into launches that fill whole tile "
                             "rounds of the chip: pipeline.vit_batch_plan).")
    parser.add_argument('--category_schedule', default="pipeline", choices=["pipeline", "sequential", "per_gpu"],
                        help="pipeline: all categories as one job (pipeline.run_categories); sequential: one category "
                             "after another, as the reference loops them; per_gpu: whole categories to GPUs.")
    parser.add_argument('--rehearse_on_one_gpu', action="store_true",
                        help="under torchrun: all ranks share cuda:0 over gloo (RCCL refuses two ranks per device) - a "
                             "rehearsal of the multi-rank run on a one-GPU box; the artefacts are those of the real run.")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    print("\n".join("%s: %s" % (k, str(v)) for k, v in sorted(dict(vars(args)).items())))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        global device
        if args.rehearse_on_one_gpu:
            device = torch.device("cuda:0")
        torch.cuda.set_device(device)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # explicit timeout: a rank that dies or hangs ends the job after two minutes instead of never
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
        else:
            dist.init_process_group(backend="nccl", device_id=device, timeout=datetime.timedelta(seconds=120))
        probe = torch.ones(1, device=device)             # RCCL smoke before any work: a 1-element all-reduce
        dist.all_reduce(probe)
        if int(probe.item()) != dist.get_world_size():
            raise RuntimeError("RCCL all-reduce smoke failed: %r" % probe.item())
    modes = ["unsupervised", "supervised"] if args.supervised == "both" else [args.supervised]
    backbone = patchcore.backbones.load(args.backbone_names[0].split(".seed-")[0])   # once, not once per category
    results = {}
    for supervised in modes:
        save_path = args.output_dir + "/" + args.dataset + "/" + args.backbone_names[0] + "/" + supervised
        os.makedirs(save_path, exist_ok=True)
        with open(os.path.join(save_path, "WEIGHTS_SOURCE.txt"), "w") as f:     # random-init runs stay recognisable
            f.write(str(getattr(backbone, "weights_source", "unknown")) + "\n")
    categories = list(args.categories or _CLASSNAMES)
    info_dir = args.output_dir + "/" + args.dataset + "/info"
    if args.category_schedule != "sequential" and len(categories) > 1:
        save_paths = {m: args.output_dir + "/" + args.dataset + "/" + args.backbone_names[0] + "/" + m for m in modes}
        return make_categories_data(
            path=args.path, categories=categories, pretrain_embed_dimension=args.pretrain_embed_dimension,
            target_embed_dimension=args.target_embed_dimension, backbone_names=args.backbone_names,
            layers_to_extract_from=args.layers_to_extract_from, patchsize=args.patchsize, save_paths=save_paths,
            train_ratio=args.train_ratio, tau=(args.tau_list if args.tau_list else args.tau), modes=modes,
            dataset=args.dataset, info_dir=info_dir, backbone=backbone, batch_size=args.batch_size,
            schedule=args.category_schedule)
    for category in categories:
        cache = {}                                       # test images + embeddings shared by the modes of a category
        for supervised in modes:
            save_path = args.output_dir + "/" + args.dataset + "/" + args.backbone_names[0] + "/" + supervised
            results[(supervised, category)] = make_category_data(
                path=args.path, category=category, pretrain_embed_dimension=args.pretrain_embed_dimension,
                target_embed_dimension=args.target_embed_dimension, backbone_names=args.backbone_names,
                layers_to_extract_from=args.layers_to_extract_from, patchsize=args.patchsize, save_path=save_path,
                tau=(args.tau_list if args.tau_list else args.tau), train_ratio=args.train_ratio,
                supervised=supervised, dataset=args.dataset,
                info_dir=info_dir, backbone=backbone,
                batch_size=args.batch_size, cache=cache)
    return results


if __name__ == "__main__":
    main()
