# gpurun -- bash scratch/ln_fusion_kernel_stats.sh : per-kernel times of a ViT-B/8 forward at 83 images (12 blocks) under
# `rocprofv3 --kernel-trace --stats`, same box, for: LayerNorm inside the residual GEMMs against separate launches.
# NEEDS A LIBRARY BUILT WITH scratch/exp_layernorm_in_residual_gemm.patch APPLIED (it adds ac_set_ln_fusion): the experiment was
# bit-equal and slower and did NOT ship (profiles/r05_vit_layernorm_in_residual_gemm_ab.txt) - the product library has no such
# entry point, and scratch/ab_ln_fusion.py stops with a message on it.  Further builds as name=lib:mode (mode 0 separate / 1 fused).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/anomaly-clustering_amd/models/acnative
for spec in "fused=$L/libac_hip.so:1" "separate=$L/libac_hip.so:0" "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; lib=${rest%%:*}; mode=${rest##*:}
  O=$R/gpurun_out/lnf_$name
  rm -rf $O
  AC_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scratch/ab_ln_fusion.py --images ${IMAGES:-83} --rounds 6 --reps 5 --only $mode > $O.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/ln_fusion_kernel_stats_$name.csv
  rm -rf $O
  echo "== $name ($(grep images $O.log | tail -1))"
  python3 - $R/gpurun_out/ln_fusion_kernel_stats_$name.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:7]:
    print("  %-64s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("  all kernels: %.2f ms" % (tot / 1e6))
PY
done
