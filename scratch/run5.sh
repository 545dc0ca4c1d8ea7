cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r2e/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e/tests.log
tail -8 gpurun_out/r2e/tests.log
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r2e/bench_$i.json 2>/dev/null; python - <<PY
import json
d=json.load(open("gpurun_out/r2e/bench_$i.json"))
print("bench", round(d["value"],1), round(d["ms_per_step"],2), "alpha ms", round(d["roofline"]["avg_launch_ms"],2), "TF", round(d["roofline"]["achieved"],1), d["stage_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r2e/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2e/pmc_$c.json 2>$R/gpurun_out/r2e/pmc_$c.err
  f=$(find $R/gpurun_out/r2e/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if r["Counter_Name"]=="$c": acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:4]:
    print("$c", k, len(v), "mean KiB %.0f -> GB %.2f" % (sum(v)/len(v), sum(v)/len(v)*1024*(2 if "$c"=="FETCH_SIZE" else 1)/1e9))
PY
  find $R/gpurun_out/r2e/pmc_$c -name "*.csv" -size +2M -delete
done
