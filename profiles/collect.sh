# Run on the GPU box: gpurun -- bash profiles/collect.sh   (bench + rocprofv3 kernel stats + one --pmc pass per counter)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final   # then: python profiles/summarize.py gpurun_out/prof_final/flat r01_bench "bottle:unsupervised:83:gpus1"
rm -rf $O; mkdir -p $O
python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_plain.json 2>$O/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 > $O/bench_under_rocprof.json 2>$O/stats.err
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 > $O/pmc_$c.json 2>$O/pmc_$c.err
done
mkdir -p $O/flat
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/flat/ \;
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do for f in $(find $O/pmc_$c -name "*counter_collection.csv"); do cp $f $O/flat/${c}_counter_collection.csv; done; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -c 600 $O/bench_plain.json; ls -la $O/flat
