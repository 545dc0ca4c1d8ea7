cd $GRAFT_REPO_ROOT
for cfg in 3 4 "5a --n 2000"; do
  for mode in naive128 plan256 naive128 plan256; do
    if [ $mode = naive128 ]; then export AC_VIT_BATCH_PLAN=naive; B=128; else unset AC_VIT_BATCH_PLAN; B=256; fi
    python bench.py --config $cfg --batch $B --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config $cfg', '$mode', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms')"
  done
done
