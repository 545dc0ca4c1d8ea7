# gpurun -- bash scratch/vit_kernel_stats.sh name=lib.so [name2=lib2.so ...]   : per-kernel times of ONE ViT block at 83 images for every
# given build of the library, each under `rocprofv3 --kernel-trace --stats` (same box, same profiler overhead; the program itself after --).
# Output: gpurun_out/vit_kernel_stats_<name>.csv and a short table on stdout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  O=$R/gpurun_out/vks_$name
  rm -rf $O
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scratch/ab_vit.py --images ${IMAGES:-83} --rounds 30 x=$lib > $O.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/vit_kernel_stats_$name.csv
  rm -rf $O
  echo "== $name"
  python3 - $R/gpurun_out/vit_kernel_stats_$name.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("  %-70s calls %5s  avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
