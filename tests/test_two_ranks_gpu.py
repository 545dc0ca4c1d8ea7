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


def _worker(rank, world, init_file, mode, out_dir):
    sys.path[:0] = [ROOT, MODELS]
    from patchcore.pipeline import HipOps, run_category
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    core = _build(dev)
    test, train = _images()
    out = run_category(HipOps(core, batch_size=4), test.to(dev), mode, tau=1.0, train_images=train.to(dev))
    np.save(os.path.join(out_dir, "alpha_%d.npy" % rank), out["alpha"].cpu().numpy())
    np.save(os.path.join(out_dir, "X_%d.npy" % rank), out["X"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["unsupervised", "supervised"])
def test_two_ranks_on_one_gpu_match_one_rank(dev, mode):
    from patchcore.pipeline import HipOps, run_category
    core = _build(dev)
    test, train = _images()
    ref = run_category(HipOps(core, batch_size=4), test.to(dev), mode, tau=1.0, train_images=train.to(dev))
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "rdzv"), mode, d), nprocs=2, join=True)
        for r in range(2):
            a = np.load(os.path.join(d, "alpha_%d.npy" % r))
            X = np.load(os.path.join(d, "X_%d.npy" % r))
            # batches of different size take different tile paths in the ViT GEMMs: fp32-level, not bitwise
            np.testing.assert_allclose(a, ref["alpha"].cpu().numpy(), rtol=2e-4, atol=1e-8)
            np.testing.assert_allclose(X, ref["X"], rtol=0, atol=2e-5)


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
