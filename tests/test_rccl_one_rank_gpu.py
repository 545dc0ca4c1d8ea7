"""GPU, ONE device: RCCL itself initialises and accepts every collective the pipeline issues, in the exact tensor
forms the pipeline passes - with a world of one rank (the only RCCL world a 1-GPU box allows; two ranks need two
devices: tests/test_rccl_gpu.py).  What this can catch before the first multi-GPU run: the library failing to load or
to create a communicator on the box, a dtype / reduce-op pair ProcessGroupNCCL rejects (int64 SUM, int32 MAX on a slice
view, fp32 MAX on a float view of an int64 buffer), all_gather_into_tensor on the padded row blocks, the object
all-gather of AC_RING_CHECK, the barrier with a bound device, the explicit timeout.  What it cannot: point-to-point
transfers (a rank cannot send to itself) and anything about bandwidth."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT, MODELS

pytestmark = pytest.mark.gpu


def test_rccl_world_of_one_accepts_the_pipelines_collectives(tmp_path):
    script = tmp_path / "one_rank.py"
    script.write_text(textwrap.dedent("""
        import datetime, os, sys
        sys.path[:0] = [%r, %r]
        import torch, torch.distributed as dist
        from patchcore.pipeline import all_gather_rows
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], HSA_ENABLE_IPC_MODE_LEGACY="0")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
        assert dist.get_backend() == "nccl"
        probe = torch.ones(1, device=dev); dist.all_reduce(probe); assert int(probe.item()) == 1      # bench.py's smoke
        acc = torch.arange(784 * 3, dtype=torch.int64, device=dev)                                    # the int64 sums
        dist.all_reduce(acc, op=dist.ReduceOp.SUM); assert int(acc[5]) == 5
        m_all = torch.zeros(4, dtype=torch.int64, device=dev).view(torch.float32)                      # maxima: fp32 view
        m_all[2] = 3.5; dist.all_reduce(m_all, op=dist.ReduceOp.MAX); assert float(m_all[2]) == 3.5
        flags32 = torch.zeros(4, dtype=torch.int64, device=dev).view(torch.int32)[:3]                  # flags: int32 slice
        flags32[1] = 1; dist.all_reduce(flags32, op=dist.ReduceOp.MAX); assert flags32.tolist() == [0, 1, 0]
        mf = torch.full((2, 6), float("inf"), device=dev); dist.all_reduce(mf, op=dist.ReduceOp.MIN)   # "slices" schedule
        send = torch.arange(12.0, device=dev).reshape(3, 4)
        recv = torch.empty(3, 4, device=dev); dist.all_gather_into_tensor(recv, send); assert torch.equal(recv, send)
        assert all_gather_rows(send, [3]) is send
        seen = [None]; dist.all_gather_object(seen, (2, 4)); assert seen == [(2, 4)]
        dist.barrier(); torch.cuda.synchronize()
        dist.destroy_process_group()
        print("rccl one rank ok")
    """ % (ROOT, MODELS)))
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script), str(port)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl one rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
