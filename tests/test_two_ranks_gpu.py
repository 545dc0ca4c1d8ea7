"""GPU: the real multi-rank code path (HipOps + sharding + collectives) with two processes that share
the one GPU of the test box.  RCCL refuses two ranks on one device, so the process group here is
gloo (it stages device tensors through the host); everything else - sharded ViT batches, the
partitioned symmetric all-pairs kernel, the MIN all-reduce, the all-gathers - is the code an 8-GPU
RCCL run executes.  Two ranks must reproduce the one-rank alpha and X."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, MODELS

pytestmark = pytest.mark.gpu


def _build(dev):
    import patchcore.common as common
    import patchcore.sampler
    import patchcore.vision_transformer as vits
    from patchcore.patchcore import AnomalyClusteringCore
    torch.manual_seed(0)
    model = vits.vit_base(patch_size=8, num_classes=0, depth=2).eval()
    for p in model.parameters():
        p.requires_grad = False
    core = AnomalyClusteringCore(dev)
    core.load(backbone=model, layers_to_extract_from=["blocks.0", "blocks.1"], device=dev, input_shape=(3, 224, 224),
              pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), nn_method=common.FaissNN(False, 4))
    return core


def _images():
    from patchcore.datasets import synthetic
    labels = ["good", "good", "crack", "crack", "hole", "hole", "good"]
    test = synthetic.make_images("bottle", labels)
    train = synthetic.make_images("bottle", ["good"] * 5, split=synthetic.DatasetSplit.TRAIN)
    return test, train


def _worker(rank, world, init_file, mode, out_dir, lean=None):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import HipOps, run_category
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    core = _build(dev)
    test, train = _images()
    ops = HipOps(core, batch_size=4)
    ops.LEAN_CHUNK = 3
    out = run_category(ops, test.to(dev), mode, tau=1.0, train_images=train.to(dev), lean=lean)
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].cpu().numpy())
    np.save(os.path.join(out_dir, "X_%d.npy" % rank), out["X"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("unsupervised", 2), ("supervised", 2), ("unsupervised", 4), ("supervised", 4)])
def test_two_ranks_on_one_gpu_match_one_rank(dev, mode, world):
    """world = 4 (shards of 2 2 2 1 images, bank 2 2 1 0): the first size at which the ring has more than one later step, so
    the fused path runs on the real kernels (round 6, pipeline.merge_later_steps / _Ring.get_run: the received taps of
    steps 1 and 2 side by side in one store, ONE aggregation + pack, one two-way rectangle - or, for the ranks whose ring
    wraps round image 0, two; supervised: the three remote bank shards as one key block, one scan)."""
    from patchcore.pipeline import HipOps, run_category
    core = _build(dev)
    test, train = _images()
    ref = run_category(HipOps(core, batch_size=4), test.to(dev), mode, tau=1.0, train_images=train.to(dev))
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, os.path.join(d, "rdzv"), mode, d), nprocs=world, join=True)
        for r in range(world):
            a = np.load(os.path.join(d, "alpha_%d.npy" % r))
            X = np.load(os.path.join(d, "X_%d.npy" % r))
            # the ring carries the ViT taps here (HipOps with a backbone): the receiver re-aggregates and packs them.
            # A ViT batch of another size gives every image the same bits (the GEMM kernels accumulate each output
            # row in one fixed K order), so two ranks equal one rank exactly
            assert np.array_equal(a, ref["alpha"].cpu().numpy()) and np.array_equal(X, ref["X"])


@pytest.mark.parametrize("mode", ["unsupervised", "supervised", "average"])
def test_lean_run_keeps_taps_instead_of_Z(dev, mode):
    """run_category(lean=True): the rank keeps the ViT taps (4.8 MB per image) instead of Z (12.85 MB); the packed blocks
    and X are rebuilt from them chunk by chunk (chunks of 3 here: 3 + 3 + 1 images) by the same kernels on the same
    bytes - alpha, X and the distances equal the run that holds Z, bit for bit, on one rank and on two."""
    from patchcore.pipeline import HipOps, Taps, run_category
    core = _build(dev)
    test, train = _images()
    ops = HipOps(core, batch_size=4)
    ops.LEAN_CHUNK = 3
    assert not ops.wants_lean(7) and ops.wants_lean(10 ** 6)
    ref = run_category(ops, test.to(dev), mode, tau=1.0, train_images=train.to(dev), lean=False)
    out = run_category(ops, test.to(dev), mode, tau=1.0, train_images=train.to(dev), lean=True)
    assert isinstance(out["Z_loc"], Taps) and out["Z_loc"].n == 7 and isinstance(ref["Z_loc"], torch.Tensor)
    again = run_category(ops, test.to(dev), mode, tau=1.0, train_images=train.to(dev), Z_loc=out["Z_loc"])
    for o in (out, again):
        assert torch.equal(o["alpha"], ref["alpha"]) and np.array_equal(o["X"], ref["X"])
        assert torch.equal(o["pdist"], ref["pdist"])
    if mode == "average":
        return
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "rdzv"), mode, d, True), nprocs=2, join=True)
        for r in range(2):
            assert np.array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].cpu().numpy())
            assert np.array_equal(np.load(os.path.join(d, "X_%d.npy" % r)), ref["X"])


def _worker_small(rank, world, init_file, out_dir):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import HipOps, run_category
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    core = _build(dev)
    test, _ = _images()
    out = run_category(HipOps(core, batch_size=4), test[:2].to(dev), "unsupervised", tau=1.0)   # 2 images, 3 ranks
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_more_ranks_than_images(dev):
    """N = 2 images over 3 ranks: one rank owns an empty shard and still takes part in every collective."""
    from patchcore.pipeline import HipOps, run_category
    core = _build(dev)
    test, _ = _images()
    ref = run_category(HipOps(core, batch_size=4), test[:2].to(dev), "unsupervised", tau=1.0)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_small, args=(3, os.path.join(d, "rdzv"), d), nprocs=3, join=True)
        for r in range(3):
            a = np.load(os.path.join(d, "alpha_%d.npy" % r))
            assert a.shape == (2, 1, 784)
            np.testing.assert_allclose(a, ref["alpha"].cpu().numpy(), rtol=2e-4, atol=1e-8)


def _worker_engine(rank, world, init_file, mode, out_dir, schedule, budget, ties=False):
    """The alpha engine alone (embeddings given), so that ranks can be compared BIT FOR BIT."""
    sys.path[:0] = [ROOT, MODELS, os.path.join(ROOT, "tests")]
    from patchcore.pipeline import HipOps, run_category
    from _synth import clustered_embeddings
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    Zs, _ = clustered_embeddings(7, 100, 256, 3, salt=51)
    Zb, _ = clustered_embeddings(5, 100, 256, 1, salt=52)
    if ties:                                   # a bank of copies of one image, one candidate slot: every list overflows
        Zb = np.repeat(Zb[:1], 6, axis=0)
        ops.nn_cap = 1
    out = run_category(ops, torch.from_numpy(Zs).to(dev), mode, tau=1.0, train_images=torch.from_numpy(Zb).to(dev),
                       schedule=schedule, budget=budget, train_ratio=1.0)
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].cpu().numpy())
    np.save(os.path.join(out_dir, "w_%d.npy" % rank), out["w_loc"].cpu().numpy())
    np.save(os.path.join(out_dir, "X_%d.npy" % rank), out["X"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,schedule,budget", [
    (2, "unsupervised", "ring", 2 * 2 * 100 * 4),      # key buffers of 2 images: streaming + sub-blocks + the split step
    (3, "unsupervised", "ring", 1 * 1 * 100 * 4),      # odd world, 1-image blocks, a short last shard
    (2, "unsupervised", "slices", 1 << 30),             # all-gathered packed shards + sliced symmetric launch
    (3, "supervised", "auto", 2 * 2 * 100 * 4),         # bank shards round the ring
])
def test_streaming_schedules_match_one_rank_bitwise(dev, world, mode, schedule, budget):
    """Two / three real ranks on one GPU with a forced tiny key-buffer budget, so that shard streaming,
    sub-blocking, symmetric halving and the int64 all-reduce all engage: alpha, w and X are BIT-EQUAL to the
    one-rank, one-launch result (the sum over key images is fixed point, so no order can change it)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from patchcore.pipeline import HipOps, run_category, shard_bounds
    from _synth import clustered_embeddings
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    Zs, _ = clustered_embeddings(7, 100, 256, 3, salt=51)
    Zb, _ = clustered_embeddings(5, 100, 256, 1, salt=52)
    ref = run_category(ops, torch.from_numpy(Zs).to(dev), mode, tau=1.0, train_images=torch.from_numpy(Zb).to(dev))
    bounds = shard_bounds(7, world)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_engine, args=(world, os.path.join(d, "rdzv"), mode, d, schedule, budget), nprocs=world, join=True)
        for r in range(world):
            assert np.array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].cpu().numpy()), r
            assert np.array_equal(np.load(os.path.join(d, "X_%d.npy" % r)), ref["X"]), r
            lo, hi = bounds[r]
            assert np.array_equal(np.load(os.path.join(d, "w_%d.npy" % r)), ref["w_loc"][lo:hi].cpu().numpy()), r


def test_pruned_search_gives_up_on_tie_heavy_banks_across_shards(dev):
    """Three ranks, a bank of six copies of one image (two per rank), one candidate slot: on its first bank shard every
    query overflows (re-measured exhaustively on the spot), so the remaining shards skip the first pass altogether.
    The answer is the exhaustive path's, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from patchcore.pipeline import HipOps, run_category, shard_bounds
    from _synth import clustered_embeddings
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    Zs, _ = clustered_embeddings(7, 100, 256, 3, salt=51)
    Zb, _ = clustered_embeddings(5, 100, 256, 1, salt=52)
    Zb = np.repeat(Zb[:1], 6, axis=0)
    ref = run_category(ops, torch.from_numpy(Zs).to(dev), "supervised", tau=1.0, train_images=torch.from_numpy(Zb).to(dev),
                       train_ratio=1.0, prune=False)
    bounds = shard_bounds(7, 3)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_engine, args=(3, os.path.join(d, "rdzv"), "supervised", d, "auto", 1 << 30, True), nprocs=3, join=True)
        for r in range(3):
            lo, hi = bounds[r]
            assert np.array_equal(np.load(os.path.join(d, "w_%d.npy" % r)), ref["w_loc"][lo:hi].cpu().numpy()), r
            assert np.array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].cpu().numpy()), r


def _worker_empty_shard_overflow(rank, world, init_file, out_dir):
    sys.path[:0] = [ROOT, MODELS, os.path.join(ROOT, "tests")]
    from patchcore.pipeline import HipOps, run_category
    from _synth import clustered_embeddings
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    ops.nn_cap = 1
    Zs, _ = clustered_embeddings(2, 100, 256, 2, salt=61)
    Zb, _ = clustered_embeddings(5, 100, 256, 1, salt=62)
    Zb = np.repeat(Zb[:1], 6, axis=0)
    out = run_category(ops, torch.from_numpy(Zs).to(dev), "supervised", tau=1.0, train_images=torch.from_numpy(Zb).to(dev),
                       train_ratio=3.0)
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_overflow_with_an_empty_query_shard(dev):
    """Round-3 advice: 2 test images over 3 ranks (rank 2 holds NO query image), a bank of six copies of one image and
    one candidate slot, so every query of ranks 0 and 1 overflows.  The flag is MAX all-reduced, so rank 2 enters the
    overflow branch too - with zero-size tensors - and must carry the bank shards round the second ring instead of
    raising (which left the other ranks waiting in batch_isend_irecv).  Result = the exhaustive one-rank path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from patchcore.pipeline import HipOps, run_category
    from _synth import clustered_embeddings
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    Zs, _ = clustered_embeddings(2, 100, 256, 2, salt=61)
    Zb, _ = clustered_embeddings(5, 100, 256, 1, salt=62)
    Zb = np.repeat(Zb[:1], 6, axis=0)
    ref = run_category(ops, torch.from_numpy(Zs).to(dev), "supervised", tau=1.0, train_images=torch.from_numpy(Zb).to(dev),
                       train_ratio=3.0, prune=False)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_empty_shard_overflow, args=(3, os.path.join(d, "rdzv"), d), nprocs=3, join=True)
        for r in range(3):
            assert np.array_equal(np.load(os.path.join(d, "alpha_%d.npy" % r)), ref["alpha"].cpu().numpy()), r


def _worker_categories_gpu(rank, world, init_file, out_dir, mode, schedule):
    sys.path[:0] = [ROOT, MODELS, os.path.join(ROOT, "tests")]
    from patchcore.pipeline import HipOps, run_categories
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    ops.embed_many = lambda sets: [x.contiguous() for x in sets]
    outs = run_categories(ops, _engine_items(dev), mode, tau=1.0, train_ratio=1.0, schedule=schedule)
    for c, o in enumerate(outs):
        np.save(os.path.join(out_dir, "alpha_%d_%d.npy" % (rank, c)), o["alpha"].cpu().numpy())
        np.save(os.path.join(out_dir, "X_%d_%d.npy" % (rank, c)), o["X"])
    dist.barrier()
    dist.destroy_process_group()


def _engine_items(dev):
    from _synth import clustered_embeddings
    items = []
    for c, (n, m) in enumerate([(7, 5), (3, 4), (5, 6)]):
        Zs, _ = clustered_embeddings(n, 100, 256, 3, salt=70 + c)
        Zb, _ = clustered_embeddings(m, 100, 256, 1, salt=80 + c)
        items.append({"images": torch.from_numpy(Zs).to(dev), "train_images": torch.from_numpy(Zb).to(dev)})
    return items


@pytest.mark.parametrize("world,mode,schedule", [(2, "unsupervised", "pipeline"), (3, "supervised", "pipeline"),
                                                 (3, "unsupervised", "per_gpu")])
def test_categories_as_one_job_match_the_loop_bitwise(dev, world, mode, schedule):
    """run_categories on the real kernels (embeddings given): the cross-category pipeline - every ring of the group in
    flight, step 0 of all categories first, ONE all-reduce of all maxima / all int64 sums / all overflow flags - and the
    category-per-GPU schedule, on 2-3 real ranks sharing the GPU, are BIT-EQUAL to run_category per category on one rank."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from patchcore.pipeline import HipOps, run_category, run_categories
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    ops.embed_many = lambda sets: [x.contiguous() for x in sets]
    items = _engine_items(dev)
    ref = [run_category(ops, it["images"], mode, tau=1.0, train_images=it["train_images"], train_ratio=1.0) for it in items]
    one = run_categories(ops, items, mode, tau=1.0, train_ratio=1.0)
    for o, r in zip(one, ref):
        assert torch.equal(o["alpha"], r["alpha"]) and np.array_equal(o["X"], r["X"])
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_categories_gpu, args=(world, os.path.join(d, "rdzv"), d, mode, schedule), nprocs=world, join=True)
        for rk in range(world):
            for c, r in enumerate(ref):
                assert np.array_equal(np.load(os.path.join(d, "alpha_%d_%d.npy" % (rk, c))), r["alpha"].cpu().numpy()), (rk, c)
                assert np.array_equal(np.load(os.path.join(d, "X_%d_%d.npy" % (rk, c))), r["X"]), (rk, c)


def test_joint_vit_batches_match_per_category_batches(dev):
    """embed_device_many: three image sets through the ViT as one stream of batches of 4 (batches span the set
    boundaries through the staging buffer) against embed_device per set."""
    core = _build(dev)
    test, train = _images()
    sets = [test[:3].to(dev), train[:5].to(dev), test[3:].to(dev), test[:0].to(dev)]
    Zs = core.embed_device_many(sets, batch_size=4)
    assert [z.shape[0] for z in Zs] == [3, 5, 4, 0]
    for s, z in zip(sets[:3], Zs):
        ref = core.embed_device(s, batch_size=4)
        # the same image in a batch of another size takes another tile path in the GEMMs only if the kernels differ in
        # their K order - they do not (tests/test_gemm_generations_gpu.py): bit-equal
        assert torch.equal(z, ref)
