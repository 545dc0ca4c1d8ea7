# Average shader clock of the 256x256 GEMM, full kernel vs the -DAC_ABLATE5 timing builds (GRBM_GUI_ACTIVE / duration).
# Needs -DAC_EXPERIMENTAL builds build_ablate/libac5_ablate{1,11}.so and libac_hip_experimental.so as `full`; run with: gpurun -- bash profiles/clock_ablation.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in full ab11 ab1; do
  case $tag in full) L=$R/anomaly-clustering_amd/models/acnative/libac_hip_experimental.so;; ab11) L=$R/build_ablate/libac5_ablate11.so;; ab1) L=$R/build_ablate/libac5_ablate1.so;; esac
  AC_LIB=$L AC_POLICY=108 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_$tag -- python3 $R/tests/perf_probe.py alpha 48 > $R/gpurun_out/clk_$tag.log 2>&1
  tail -1 $R/gpurun_out/clk_$tag.log
done
