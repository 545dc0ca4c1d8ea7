cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2g/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g/tests.log
tail -4 gpurun_out/r2g/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in 2-pill 3 4; do timeout 900 python bench.py --config $c --no-cpu-baseline > gpurun_out/r2g/bench_$c.json 2>/dev/null; done
timeout 900 python bench.py --config 5a --n 2000 --no-cpu-baseline > gpurun_out/r2g/bench_5a_n2000.json 2>/dev/null
timeout 900 python bench.py --config 5b --n 2000 --no-cpu-baseline > gpurun_out/r2g/bench_5b_n2000.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2g/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms", "roof", round(d["roofline"]["achieved"],1), round(d["roofline"]["frac"],3), "ward", d.get("ward_metrics_ms"))
    except Exception as e:
        print(f, "FAILED", e)
PY
