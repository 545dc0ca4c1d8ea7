# where the joules of the all-pairs kernel go: the k loop with parts removed (-DAC_ABLATE builds; results are wrong, timing
# and power are what is read).  1 = no DMA in the loop, 2 = no fragment reads after the first stage, 4 = no MFMA.
cd $GRAFT_REPO_ROOT
L=anomaly-clustering_amd/models/acnative
for v in "" _abl1 _abl2 _abl3 _abl4 _abl6 _abl7; do
  echo "== libac_hip$v.so"
  python scratch/power_probe.py --seconds 3 --only-real --lib $L/libac_hip$v.so 2>&1 | grep -v "amdgpu.ids\|^idle" | cut -c1-150
done
