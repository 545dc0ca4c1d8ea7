# Run on the GPU box:  gpurun --timeout 1200 -- bash profiles/collect_r06.sh [config] [tag] [steps]
# bench + rocprofv3 kernel stats + PMC passes (HBM bytes, clock, MFMA pipe) + the stall-attribution SQ passes of
# VERDICT r02 item 3.  Every rocprofv3 command has the program itself after `--`; --pmc passes carry --kernel-trace only.
# Then here:  python profiles/summarize.py gpurun_out/prof_r06_<tag>/flat r06_<tag> "<workload key>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-2}
TAG=${2:-bench}
STEPS=${3:-5}
O=$R/gpurun_out/prof_r06_$TAG
rm -rf $O; mkdir -p $O/flat   # (merge note: delete the local gpurun_out/prof_* directory before re-running, or stale *_kernel_stats.csv files of earlier runs stay beside the new one)
B="python3 $R/bench.py --config $CFG"
timeout 900 $B --steps $STEPS --warmup 1 > $O/bench_plain.json 2>$O/bench_plain.err || exit 1
echo "plain bench done"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps $STEPS --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/stats.err || exit 1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/flat/ \;
rm -rf $O/stats
echo "kernel stats done"
rocprofv3 -L > $O/counters_available.txt 2>&1
pass() {   # pass <name> <counter...>: one --pmc pass with the counters this box lists
  name=$1; shift
  have=""
  for c in "$@"; do if grep -qw "$c" $O/counters_available.txt; then have="$have $c"; else echo "counter $c not listed" >> $O/missing.txt; fi; done
  [ -z "$have" ] && return 0
  timeout 600 rocprofv3 --pmc $have --kernel-trace --output-format csv -d $O/pmc_$name -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$name.json 2>$O/pmc_$name.err
  rc=$?
  for f in $(find $O/pmc_$name -name "*counter_collection.csv"); do cp $f $O/flat/${name}_counter_collection.csv; done
  rm -rf $O/pmc_$name
  echo "pmc pass $name ($have) rc=$rc"
  return $rc
}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass clock GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES || exit 1
if [ "$CFG" = "2" ]; then
  pass waits SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM || exit 1
  pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT || exit 1
  pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16 || exit 1
  pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES || exit 1
  pass l2 TCC_HIT_sum TCC_MISS_sum || exit 1
fi
tail -c 400 $O/bench_plain.json; echo; ls $O/flat; cat $O/missing.txt 2>/dev/null; exit 0
