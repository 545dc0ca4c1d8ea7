# gpurun -- bash scratch/rank_trace.sh [G]: kernel trace of rank 0's compute of a G-rank run of config 2 (scratch/predict_scaling.py,
# one GPU): the sum of the kernel times against the measured wall time says how much of a small rank's step is launch gaps.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=${1:-8}
O=$R/gpurun_out/rank_trace_g$G
rm -rf $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scratch/predict_scaling.py --categories bottle --gpus $G --reps 5 > $O.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/rank_trace_g${G}_kernel_stats.csv
t=$(find $O -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && cp $t $R/gpurun_out/rank_trace_g${G}_kernel_trace.csv
rm -rf $O
tail -30 $O.log | cut -c1-300
