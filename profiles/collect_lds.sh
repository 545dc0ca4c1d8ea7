cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_lds
rm -rf $O; mkdir -p $O
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$c.json 2>$O/pmc_$c.err
  for f in $(find $O/pmc_$c -name "*counter_collection.csv"); do cp $f $O/${c}_counter_collection.csv; done
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O
