cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 1200 python -m pytest tests -m gpu -q -s -k "dino_like or full_size or config1 or both_modes or main_then_test" 2>&1 | grep -E "max \||err|NMI|passed|failed|N=|truth" > gpurun_out/r2d/numbers.log
cat gpurun_out/r2d/numbers.log
for c in 3 4; do timeout 900 python bench.py --config $c --no-cpu-baseline > gpurun_out/r2d/bench_c$c.json 2> gpurun_out/r2d/bench_c$c.err; python - <<PY
import json
d=json.load(open("gpurun_out/r2d/bench_c$c.json"))
print("config $c", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["launches_per_step"], d.get("h2d_inclusive"), d.get("ward_metrics_ms"), d.get("stage_ms"))
print([ (p["category"], p["n"], round(p["ms"],1)) for p in d["per_category"]])
PY
done
timeout 600 python bench.py --config 2-pill --no-cpu-baseline > gpurun_out/r2d/bench_pill.json 2>/dev/null; head -c 300 gpurun_out/r2d/bench_pill.json; echo
timeout 900 python bench.py --config 5a --n 2000 --no-cpu-baseline > gpurun_out/r2d/bench_5a_2000.json 2>/dev/null; head -c 300 gpurun_out/r2d/bench_5a_2000.json; echo
timeout 900 python bench.py --config 5b --n 2000 --no-cpu-baseline > gpurun_out/r2d/bench_5b_2000.json 2>gpurun_out/r2d/bench_5b.err; head -c 300 gpurun_out/r2d/bench_5b_2000.json; echo; tail -2 gpurun_out/r2d/bench_5b.err
