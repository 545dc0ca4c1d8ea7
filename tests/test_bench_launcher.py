"""CPU: `bench.py --gpus N` without a torchrun environment starts its own N ranks.  The launcher is driven here
with stub children (no GPU, no torch in the child): the environment every rank gets, the relay of rank 0's JSON
line, a failing rank, and the refusal to run on fewer devices than asked."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ac_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stub(tmp_path, body):
    p = tmp_path / "stub_rank.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_launcher_sets_rank_environment_and_relays_rank0(tmp_path, capfd):
    bench = _bench_module()
    cmd = _stub(tmp_path, """
        import json, os, sys
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]
        rec = {k: os.environ.get(k) for k in keys}
        open(os.path.join(sys.argv[1], "env%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
        if rec["RANK"] == "0":
            print("a progress note that is not JSON")
            print(json.dumps({"metric": "stub", "n_gpus": int(rec["WORLD_SIZE"])}))
        else:
            print("rank %s chatter" % rec["RANK"])          # must not reach the launcher's stdout
    """) + [str(tmp_path)]
    rc = bench.launch_ranks(3, cmd, visible_devices=3)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "n_gpus": 3}
    envs = [json.load(open(tmp_path / ("env%d.json" % r))) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_launcher_fails_when_a_rank_fails_and_ends_the_others(tmp_path, capfd):
    bench = _bench_module()
    cmd = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)                                       # a rank waiting in a collective for the dead one
    """)
    import time
    t0 = time.time()
    rc = bench.launch_ranks(2, cmd, visible_devices=2)
    out, err = capfd.readouterr()
    assert rc == 7 and time.time() - t0 < 60
    assert out.strip() == "" and "rank 1 exited with code 7" in err


def test_launcher_ends_hung_ranks_at_its_deadline(tmp_path, capfd):
    """Every rank alive and waiting (a hung collective): no child ever exits, so only the wall-clock limit can end the
    run - non-zero, within the limit, children ended by PID."""
    bench = _bench_module()
    pids = tmp_path / "pids"
    pids.mkdir()
    cmd = _stub(tmp_path, """
        import os, sys, time
        open(os.path.join(sys.argv[1], os.environ["RANK"]), "w").write(str(os.getpid()))
        time.sleep(600)
    """) + [str(pids)]
    import time
    t0 = time.time()
    rc = bench.launch_ranks(2, cmd, visible_devices=2, timeout_s=3.0)
    out, err = capfd.readouterr()
    assert rc == 124 and time.time() - t0 < 30
    assert out.strip() == "" and "still running after 3 s" in err
    for r in ("0", "1"):
        pid = int((pids / r).read_text())
        with pytest.raises(OSError):
            os.kill(pid, 0)                                   # the child is gone


def test_visible_gpus_reads_no_runtime(monkeypatch):
    """The launcher's device count comes from the KFD topology in /sys and the *_VISIBLE_DEVICES variables - it never
    loads the HIP runtime.  This container has no KFD: 0."""
    bench = _bench_module()
    n = bench.visible_gpus()
    assert isinstance(n, int) and n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0


def test_launcher_refuses_fewer_devices_than_ranks(tmp_path, capfd):
    bench = _bench_module()
    marker = tmp_path / "started"
    cmd = _stub(tmp_path, "open(%r, 'w').write('x')" % str(marker))
    assert bench.launch_ranks(2, cmd, visible_devices=1) != 0
    out, err = capfd.readouterr()
    assert out.strip() == "" and "only 1 GPU(s) visible" in err and not marker.exists()


def test_bench_cli_gpus_2_without_devices_exits_nonzero():
    """The real command line on a box with fewer than two GPUs (this container has none): non-zero, a clear
    message, nothing on stdout."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs are visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "GPU(s) visible" in r.stderr


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
