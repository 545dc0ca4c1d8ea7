"""GPU, TWO OR MORE MI355X: the first executions of the RCCL path.  Every test here is skipped on a 1-GPU box (RCCL
refuses two ranks on one device; the same code runs there over gloo, tests/test_two_ranks_gpu.py) and runs by itself
the moment a larger box appears:

  * `bench.py --gpus 2` as the driver calls it (self-launched ranks, backend nccl, RCCL smoke, config 2),
  * 2-rank `nccl` run_category: device-tensor point-to-point ring (`_Ring` with stage == False), the fp32 MAX / int64
    SUM / int32 MAX all-reduces and the all-gathers - alpha, w and X BIT-EQUAL to one rank, both modes,
  * 2-rank `nccl` run_categories (cross-category pipeline: every ring posted up front), bit-equal to the loop.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, MODELS

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")]


def test_bench_two_gpus_over_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--launch-timeout", "420"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["config"]["images_per_step"] == 83


def _engine_inputs():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _synth import clustered_embeddings
    Zs, _ = clustered_embeddings(7, 100, 256, 3, salt=51)
    Zb, _ = clustered_embeddings(5, 100, 256, 1, salt=52)
    return torch.from_numpy(Zs), torch.from_numpy(Zb)


def _engine_ops():
    from patchcore.pipeline import HipOps
    ops = HipOps(core=None)
    ops.embed = lambda images: images.contiguous()
    ops.embed_many = lambda sets: [x.contiguous() for x in sets]
    return ops


def _worker_nccl(rank, world, port, out_dir, mode, many):
    sys.path[:0] = [ROOT, MODELS, os.path.join(ROOT, "tests")]
    import datetime
    from patchcore.pipeline import run_category, run_categories
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))
    Zs, Zb = _engine_inputs()
    ops = _engine_ops()
    if many:
        items = [{"images": Zs.to(dev), "train_images": Zb.to(dev)}, {"images": Zs[:3].to(dev), "train_images": Zb[:4].to(dev)}]
        outs = run_categories(ops, items, mode, tau=1.0, train_ratio=1.0, budget=2 * 2 * 100 * 4)
    else:
        outs = [run_category(ops, Zs.to(dev), mode, tau=1.0, train_images=Zb.to(dev), train_ratio=1.0, budget=2 * 2 * 100 * 4)]
    for c, o in enumerate(outs):
        np.save(os.path.join(out_dir, "alpha_%d_%d.npy" % (rank, c)), o["alpha"].cpu().numpy())
        np.save(os.path.join(out_dir, "X_%d_%d.npy" % (rank, c)), o["X"])
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode,many", [("unsupervised", False), ("supervised", False), ("unsupervised", True), ("supervised", True)])
def test_two_rccl_ranks_are_bit_equal_to_one_rank(dev, mode, many):
    from patchcore.pipeline import run_category
    Zs, Zb = _engine_inputs()
    ops = _engine_ops()
    sets = [(Zs, Zb), (Zs[:3], Zb[:4])] if many else [(Zs, Zb)]
    ref = [run_category(ops, a.to(dev), mode, tau=1.0, train_images=b.to(dev), train_ratio=1.0) for a, b in sets]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_nccl, args=(2, _free_port(), d, mode, many), nprocs=2, join=True)
        for rk in range(2):
            for c, r in enumerate(ref):
                assert np.array_equal(np.load(os.path.join(d, "alpha_%d_%d.npy" % (rk, c))), r["alpha"].cpu().numpy()), (rk, c)
                assert np.array_equal(np.load(os.path.join(d, "X_%d_%d.npy" % (rk, c))), r["X"]), (rk, c)
