cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2h -- python3 $R/tests/perf_probe.py agg 83 > /dev/null 2>&1
f=$(find $R/gpurun_out/r2h -name "*kernel_stats.csv" | head -1)
cut -c1-150 $f | head -8
find $R/gpurun_out/r2h -name "*kernel_trace.csv" -delete
