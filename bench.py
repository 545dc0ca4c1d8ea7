#!/usr/bin/env python3
"""End-to-end images/sec of the anomaly-clustering hot path on N MI355X of one node.

A *step* = one MVTec-AD-shaped category through the whole hot path with the images already in
HBM:  ViT-B/8 extract (blocks.10/11) -> patch aggregation -> unsupervised alpha (tau=1) ->
weighted-average embeddings -> (RCCL all-gather) -> N x N pairwise distance.  Because alpha is
O(N^2), every images/sec figure is tied to the category size; the default is BASELINE.json
configs[1] with the "bottle" test set (N=83) and random-init ViT-B/8 weights (no checkpoint
or dataset is reachable offline).  With --gpus G the same category is sharded over G ranks
(strong scaling; Z and X are all-gathered over RCCL).

    python bench.py                      # 1 GPU, default steps
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel (gemm2::kernel<EpiPairMin, true>, the fused all-pairs
                 min-distance MFMA GEMM): algorithmic FLOPs per launch / average launch
                 duration measured with events on the launch stream inside the timed region.
  cpu_baseline - the CPU oracle (a port of the reference's arithmetic) timed on this host on a
                 bounded sample of the same workload, N=1 run only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "anomaly-clustering_amd", "models"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP16_MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--category", default="bottle", help="MVTec-AD category whose test-set shape is used")
    ap.add_argument("--batch", type=int, default=128, help="images per ViT launch")
    ap.add_argument("--mode", default="unsupervised", choices=["unsupervised", "supervised", "average"])
    ap.add_argument("--tau", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=6, help="images in the CPU-baseline sample")
    return ap.parse_args()


def pmc_traffic(workload_key, kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary of the SAME workload
    (profiles/*_pmc.json, written by profiles/summarize.py from separate rocprofv3 --pmc passes of this
    script; FETCH_SIZE x2 correction for gfx950).  None when no matching profile is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("_meta", {}).get("workload_key") != workload_key:
            continue
        for k, v in d.items():
            if kernel_substr in k and "hbm_read_bytes_per_launch" in v and "hbm_write_bytes_per_launch" in v:
                best = v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]
    return best


def cpu_baseline(n_sample, n_full, category):
    """Time the oracle on a bounded sample: n_sample images embedded, all ordered pairs among them
    for alpha; extrapolate to the category size by per-image and per-pair cost."""
    from oracle import vit as ovit, aggregate as oagg, alpha as oalpha, cluster as ocluster
    import patchcore.vision_transformer as vits
    from patchcore.datasets import synthetic
    torch.manual_seed(0)
    model = vits.vit_base(patch_size=8, num_classes=0)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    labels = synthetic.category_labels(category)[:n_sample]
    images = synthetic.make_images(category, labels)
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    Zs = []
    for i in range(n_sample):                                # reference runs batch size 1 (main.py:213)
        taps = ovit.forward_taps(sd, images[i:i + 1], taps=(10, 11))
        Zs.append(oagg.aggregate([taps[10], taps[11]], 3, 2048, 4096)[0])
    Z = torch.stack(Zs)
    t_embed = (time.perf_counter() - t0) / n_sample
    t0 = time.perf_counter()
    a64 = oalpha.matrix_alpha_unsupervised(1.0, 1, Z)
    t_pair = (time.perf_counter() - t0) / (n_sample * (n_sample - 1))
    t0 = time.perf_counter()
    _, X = oalpha.weighted_average(a64, Z)
    ocluster.pdist(X)
    t_tail = (time.perf_counter() - t0) / n_sample
    total = n_full * (t_embed + t_tail) + n_full * (n_full - 1) * t_pair
    return {"value": n_full / total, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d images embedded (ViT+aggregate, batch 1) + %d ordered alpha pairs on the host; "
                      "extrapolated to N=%d with %.3f s/image + %.4f s/pair"
                      % (n_sample, n_sample * (n_sample - 1), n_full, t_embed + t_tail, t_pair)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import acnative
    import patchcore.backbones
    import patchcore.common as common
    import patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from patchcore.pipeline import HipOps, run_category
    from patchcore.datasets import synthetic

    os.environ.setdefault("AC_BACKBONE_SEED", "0")       # random-init weights of the named architecture (no checkpoint offline)
    backbone = patchcore.backbones.load("dino_vitbase8")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=backbone, layers_to_extract_from=["blocks.10", "blocks.11"], device=dev,
              input_shape=(3, 224, 224), pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), anomaly_scorer_num_nn=5,
              nn_method=common.FaissNN(True, 4))
    ops = HipOps(core, batch_size=args.batch)
    labels = synthetic.category_labels(args.category)
    n = len(labels)
    images = synthetic.make_images(args.category, labels).to(dev)      # resident in HBM before timing
    train = None
    if args.mode == "supervised":
        train = synthetic.make_images(args.category, synthetic.category_labels(args.category, synthetic.DatasetSplit.TRAIN),
                                      split=synthetic.DatasetSplit.TRAIN).to(dev)

    def step(timings=None):
        return run_category(ops, images, args.mode, tau=args.tau, train_images=train, want_host=False,
                            timings=timings)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    acnative.kernel_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    events, acnative.kernel_events = acnative.kernel_events, None
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = tmax.item()

    # roofline of the dominant kernel from the events recorded inside the timed region
    roof = None
    if events:
        ms = [e0.elapsed_time(e1) for (_, e0, e1, *_s) in events]
        flops = [2.0 * (nk * P) * (qn * P) * D for (_, _, _, nk, qn, P, D) in events]
        avg_ms = sum(ms) / len(ms)
        avg_flop = sum(flops) / len(flops)
        ach = avg_flop / (avg_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "gemm2::kernel<EpiPairMin, true>", "achieved": ach,
                "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP16_MFMA_PEAK_TFLOPS,
                "traffic": pmc_traffic("%s:%s:%d:gpus%d" % (args.category, args.mode, n, world), "EpiPairMin"),
                "avg_launch_ms": avg_ms, "launches": len(ms),
                "note": "achieved = algorithmic 2*P^2*D FLOP per ORDERED image pair / launch time. The kernel issues 3 "
                        "fp16 MFMA passes per product (split-fp16, fp32-grade) and, when keys == queries, computes "
                        "only the upper triangle of tiles (each reduced both ways), so issued MFMA work = 1.5x the "
                        "algorithmic FLOPs on one GPU and frac <= 2/3 by construction; traffic = L2-miss (fabric) "
                        "bytes per launch from the committed PMC passes of this workload"}

    # per-stage breakdown (one extra, untimed-for-the-metric step)
    stages = {}
    step(timings=stages)

    if rank == 0:
        out = {
            "metric": "images/sec end-to-end (extract+alpha+pairwise)",
            "value": n * args.steps / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32 (split-fp16 MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "MVTec-AD single category (%s-shaped test set, N=%d), dino_vitbase8 "
                                   "blocks.10+11, 2048->4096, patchsize 3, %s alpha tau=%g, random-init weights"
                                   % (args.category, n, args.mode, args.tau),
                       "images_per_category": n, "vit_batch": args.batch,
                       "parallelism": "images sharded over %d rank(s), RCCL all-gather of Z and X" % world},
            "stage_ms": {k: 1e3 * v for k, v in stages.items()},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, n, args.category)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
