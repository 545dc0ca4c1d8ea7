cd $GRAFT_REPO_ROOT
export AC_LIB=$GRAFT_REPO_ROOT/build_ablate/libac_raster_ab.so
for rep in 1 2; do
for g in -4 4 -8 8 2 3 6 16; do
  echo -n "G=$g: "; AC_EXP_G=$g python tests/perf_probe.py alpha 83 2>/dev/null | tail -1
done; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in -4 4 8 2; do
  AC_EXP_G=$g rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2f/pmc_$g -- python3 $R/tests/perf_probe.py alpha 83 > /dev/null 2>&1
  f=$(find $R/gpurun_out/r2f/pmc_$g -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if r["Counter_Name"]=="FETCH_SIZE" and "EpiPairMin" in r["Kernel_Name"]: acc["pair"].append(float(r["Counter_Value"]))
print("G=$g FETCH GB per launch:", [round(v*1024*2/1e9,1) for v in acc["pair"]])
PY
  rm -rf $R/gpurun_out/r2f/pmc_$g
done
