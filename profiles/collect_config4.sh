# gpurun -- bash profiles/collect_config4.sh : rocprofv3 kernel stats of bench.py --config 4 (supervised, pruned exact search)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02_c4
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --config 4 --no-cpu-baseline --no-extras > $O/bench.json 2>$O/err.txt
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --config 4 --no-cpu-baseline --no-extras --warmup 0 > /dev/null 2>$O/pmc_$c.err
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"].split("(")[0]
    if "EpiNNCand" in k or "nn_refine" in k or "EpiPairMin" in k:
        a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[2] += 1
for k, (v, us, n) in acc.items():
    print("$c", k[:60], "launches", n, "sum_counter %.4g" % v, "sum_us %.1f" % us)
PY
done
rm -rf $O/stats $O/pmc_*; find $O -name "*.csv" -size +1M -delete
head -6 $O/kernel_stats.csv | cut -c1-170
