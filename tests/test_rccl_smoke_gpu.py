"""GPU: RCCL itself, on the one GPU a test box has.

Every multi-rank test of this repo runs over gloo (RCCL refuses two ranks on one device), so this file makes
sure that at least the RCCL side of what the multi-GPU engine calls has executed on the hardware before the
8-GPU bench does: a world-size-1 `nccl` process group (backend "nccl" IS RCCL on ROCm) created the way bench.py /
examples/main.py create theirs, and on it every collective + dtype + reduction the pipeline uses -
all_reduce SUM on int64 (the fixed-point sums), MAX on float32 (the scale), MIN on float32 (the slices schedule),
all_gather_into_tensor on uint8 and float32 (packed shards, X, alpha), a batch_isend_irecv of uint8 + float32
tensors (the ring; to itself), barrier - with values checked.  Runs in a child process: a process group cannot be
re-created inside the pytest process.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
a = torch.arange(10, dtype=torch.int64, device=dev) * (1 << 40)
dist.all_reduce(a, op=dist.ReduceOp.SUM); assert a[9].item() == 9 * (1 << 40)
m = torch.tensor([3.5], device=dev); dist.all_reduce(m, op=dist.ReduceOp.MAX); assert m.item() == 3.5
f = torch.full((4, 1000), float("inf"), device=dev); f[1, 7] = 2.0
dist.all_reduce(f, op=dist.ReduceOp.MIN); assert f[1, 7].item() == 2.0 and torch.isinf(f[0, 0])
u = torch.arange(4096, dtype=torch.int32, device=dev).to(torch.uint8).reshape(4, 1024)
out = torch.empty_like(u); dist.all_gather_into_tensor(out, u); assert torch.equal(out, u)
x = torch.randn(5, 4096, device=dev); xo = torch.empty_like(x); dist.all_gather_into_tensor(xo, x); assert torch.equal(xo, x)
ru, rx = torch.empty_like(u), torch.empty(5, dtype=torch.float32, device=dev)
sx = torch.arange(5, dtype=torch.float32, device=dev)
works = dist.batch_isend_irecv([dist.P2POp(dist.isend, u, 0), dist.P2POp(dist.isend, sx, 0),
                                dist.P2POp(dist.irecv, ru, 0), dist.P2POp(dist.irecv, rx, 0)])
for w in works: w.wait()
torch.cuda.synchronize()
assert torch.equal(ru, u) and torch.equal(rx, sx)
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_SMOKE_OK")
"""


def test_rccl_world_of_one_runs_the_collectives_the_engine_uses():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SMOKE_OK" in r.stdout, "RCCL smoke failed:\n%s\n%s" % (r.stdout[-2000:], r.stderr[-3000:])
