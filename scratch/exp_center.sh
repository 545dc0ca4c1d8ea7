set -x
cd $GRAFT_REPO_ROOT
for g in default 8; do
  for w in synth real; do
    if [ $g = default ]; then python scratch/exp_center.py $w; else AC_GEMM=$g python scratch/exp_center.py $w; fi
  done
done 2>&1 | grep -v Warning | tee gpurun_out/exp_center.log
