"""GPU: bench.py prints ONE JSON line with the contract's fields, for the workloads that can be shrunk to seconds
(config 5a / 5b at a tiny --n with a tiny minbuf budget, so that the sub-blocked schedule is what runs).  The
full-size default run is the driver's job; this guards the plumbing around it."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


_RESULTS = {}       # one bench.py run per distinct command line and session (the rehearsals compare with runs other tests make)


def _bench(*args):
    if args not in _RESULTS:
        _RESULTS[args] = _bench_run(*args)
    return _RESULTS[args]


def _bench_run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


def test_bench_config_5b_small():
    d = _bench("--config", "5b", "--n", "40", "--budget-mb", "1", "--no-cpu-baseline", "--steps", "1")
    assert REQUIRED <= set(d) and d["unit"] == "images/sec" and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert d["config"]["images_per_step"] == 40 and "N=40" in d["config"]["workload"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["peak"] == 2500.0 and 0 < roof["frac"] < 2 / 3
    assert roof["launches_per_step"] > 1                     # 1 MiB budget: 18-image blocks -> several launches
    # N(N-1) ordered pairs x 2 P^2 D per step, whatever the blocking
    total = roof["algorithmic_flop_per_launch"] * roof["launches"]
    assert abs(total - 40 * 39 * 2.0 * 784 * 784 * 4096) / total < 1e-9
    assert d["ward_metrics_ms"] > 0 and 0 <= d["clustering"]["NMI"] <= 1


def test_bench_config_5a_small():
    d = _bench("--config", "5a", "--n", "24", "--no-cpu-baseline", "--steps", "1")
    assert REQUIRED <= set(d) and d["roofline"]["kernel"].startswith("ViT forward") and d["roofline"]["frac"] < 1 / 3


def _check_all15(d, mode):
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["unit"] == "images/sec" and d["scaling"] == "strong"
    assert d["config"]["images_per_step"] == 1725 and d["config"]["categories"] == 15
    per = d["per_category"]
    assert [c["n"] for c in per] == [83, 150, 132, 117, 78, 110, 124, 115, 167, 160, 117, 42, 100, 79, 151]
    if d["config"]["category_schedule"] == "sequential":      # one clock per category only when they run one after another
        assert all(c["ms"] > 0 and c["images_per_sec"] > 0 for c in per)
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["peak"] == 2500.0 and roof["launches_per_step"] == 15
    # every ordered image pair of every category (bank: N test x M = N bank images) resolved exactly once per step
    pairs = sum(n * (n - 1.0) if mode == "unsupervised" else n * 1.0 * n for n in (c["n"] for c in per))
    total = roof["algorithmic_flop_per_launch"] * roof["launches"] / d["steps"]
    assert abs(total - pairs * 2.0 * 784 * 784 * 4096) / total < 1e-9
    assert 0 < roof["frac"] < (2 / 3 if mode == "unsupervised" else 1.0)
    # the committed PMC summary of THIS workload (profiles/r03_config*_pmc.json) supplies the HBM-side bytes
    assert roof["traffic"] is not None and roof["traffic"] > 0 and roof["traffic_source"].startswith("profiles/")
    assert set(d["stage_ms"]) >= {"embed", "pair_min_dist", "pdist"}
    assert "pinned host memory" in d["config"]["input"]       # `value` starts at the H2D copy


@pytest.mark.parametrize("schedule", ["pipeline", "sequential", "per_gpu"])
def test_bench_config_3_all_fifteen_categories(schedule):
    """BASELINE config 3 end to end through bench.py on one GPU (the 8-GPU run is the driver's): ~5 s.  The 15 categories
    as one job (default), one after another (the reference's loop), whole categories per GPU (one GPU: all on it)."""
    d = _bench("--config", "3", "--gpus", "1", "--no-cpu-baseline", "--steps", "1", "--warmup", "0", "--category-schedule", schedule)
    assert d["config"]["category_schedule"] == schedule
    _check_all15(d, "unsupervised")


def test_bench_config_4_supervised():
    _check_all15(_bench("--config", "4", "--gpus", "1", "--no-cpu-baseline", "--steps", "1", "--warmup", "0"), "supervised")


@pytest.mark.parametrize("ranks,extra", [(4, ()), (2, ("--config", "3"))])
def test_bench_rehearsal_of_the_multi_rank_run_on_one_gpu(ranks, extra):
    """`bench.py --gpus N` as the driver will call it on the 8-GPU node - the launcher, N fresh rank processes, the process
    group, the sharded step, the barrier + max-over-ranks clock, rank 0's single JSON line - rehearsed on the ONE GPU of
    this box: the ranks share the device over gloo (--rehearse-on-one-gpu; RCCL refuses two ranks per device).  The
    clustering scores of the step must be those of the one-rank run (alpha and X are bit-equal across rank counts)."""
    common = ("--no-cpu-baseline", "--steps", "1", "--warmup", "0") + tuple(extra)
    # the one-rank run it is compared with: config 3's is the run test_bench_config_3[pipeline] makes (the scores of a step do
    # not depend on warm-up), config 2's is made here
    one = (_bench("--config", "3", "--gpus", "1", "--no-cpu-baseline", "--steps", "1", "--warmup", "0", "--category-schedule", "pipeline")
           if extra else _bench("--gpus", "1", *common))
    d = _bench("--gpus", str(ranks), "--rehearse-on-one-gpu", *common)
    assert REQUIRED <= set(d) and d["n_gpus"] == ranks and "rehearsal" in d and "rehearsal" not in one
    assert d["value"] > 0 and d["config"]["images_per_step"] == one["config"]["images_per_step"]
    assert d["clustering"]["NMI"] == one["clustering"]["NMI"] and d["clustering"]["ARI"] == one["clustering"]["ARI"]
    assert d["clustering"]["F1"] == one["clustering"]["F1"]
    assert "cpu_baseline" not in d
