# gpurun -- bash scratch/pmc_pairmin.sh <tag> <lib> [N]   : L2 hit rate, fabric bytes, clock and MFMA-pipe occupancy of the
# symmetric all-pairs launch (PM_ARGS="--two-way Q": a two-way block of N x Q images instead) of ONE build of the library (scratch/ab_pairmin.py as the program; --pmc passes carry
# --kernel-trace only, the program itself after `--`).  Output: gpurun_out/pmc_pairmin_<tag>/*.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; LIB=$2; N=${3:-83}
O=$R/gpurun_out/pmc_pairmin_$TAG
rm -rf $O; mkdir -p $O
for p in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  name=$(echo $p | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/$name -- python3 $R/scratch/ab_pairmin.py --n $N ${PM_ARGS:-} --rounds 3 --no-vendor x=$LIB > $O/$name.log 2>&1
  f=$(find $O/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $O/$name.csv
  rm -rf $O/$name
done
python3 - $O <<'PY'
import csv, sys, glob, collections
o = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(o + "/*.csv"):
    for r in csv.DictReader(open(f)):
        if "EpiPairMin" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(v) / len(v) for k, v in acc.items()}
out = {"dur_us": m.get("dur_us")}
if "FETCH_SIZE" in m: out["fabric_read_GB_per_launch"] = 2 * m["FETCH_SIZE"] * 1024 / 1e9
if "TCC_HIT_sum" in m: out["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
if "GRBM_GUI_ACTIVE" in m:
    out["clock_mhz"] = m["GRBM_GUI_ACTIVE"] / 8 / m["dur_us"]
    out["mfma_pipe_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
    out["parked"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]; out["issue_stalled"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum"):
    if k in m: out[k] = m[k]
print(o.split("_")[-1], out)
PY
