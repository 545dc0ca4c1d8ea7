cd $GRAFT_REPO_ROOT
export AC_LIB=$GRAFT_REPO_ROOT/build_ablate/libac_raster_ab.so
for rep in 1 2 3; do
for g in -4 4 6 -6; do
  echo -n "G=$g: "; AC_EXP_G=$g timeout 120 python tests/perf_probe.py alpha 83 2>/dev/null | tail -1
done; done
