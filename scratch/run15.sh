cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 900 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r2i/bench_4.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2i/bench_4.json"))
print("config 4", round(d["value"],1), "img/s", round(d["ms_per_step"],1), "ms  roof", round(d["roofline"]["achieved"],1), round(d["roofline"]["frac"],3), d["roofline"]["kernel"][:40], d["stage_ms"])
PY
