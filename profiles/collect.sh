# Run on the GPU box: gpurun -- bash profiles/collect.sh   (bench + rocprofv3 kernel stats + one --pmc pass per counter)
# then here:  python profiles/summarize.py gpurun_out/prof_r02/flat r02_bench "bottle:unsupervised:83:gpus1"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
timeout 600 python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_plain.json 2>$O/bench_plain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/stats.err
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$c.json 2>$O/pmc_$c.err
done
mkdir -p $O/flat
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/flat/ \;
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do for f in $(find $O/pmc_$c -name "*counter_collection.csv"); do cp $f $O/flat/${c}_counter_collection.csv; done; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_GRBM_GUI_ACTIVE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/stats
tail -c 600 $O/bench_plain.json; ls -la $O/flat
