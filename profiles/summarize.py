#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python profiles/summarize.py <dir with *_kernel_stats.csv / *_counter_collection.csv> <round tag> [workload key]

Writes profiles/<tag>_kernel_stats.csv (top kernels, copied verbatim columns) and
profiles/<tag>_pmc.json: per kernel the mean of every collected counter, the mean duration, and
for FETCH_SIZE / WRITE_SIZE the HBM-side bytes per launch with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request: x2 for wide coalesced streams; both
counters are in KiB).

_meta of the PMC summary says what the numbers belong to: `workload_key` (bench.py matches on it), `round` (from the tag),
`collected_at` (UTC time stamp of the newest counter file = when the box wrote it) and `head` - the commit checked out HERE when
the summary is written, "-dirty" when tracked files differ from it.  Collect from a committed tree and summarise before
the next edit, and `head` is the code the box ran; bench.py prints it next to roofline.traffic (traffic_head), and picks the
newest matching profile by collected_at, not by file name."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time


def git_head(root):
    try:
        h = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        d = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True,
                           timeout=10)
        return (h.stdout.strip() + ("-dirty" if d.stdout.strip() else "")) if h.returncode == 0 else None
    except Exception:
        return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out_dir = os.path.dirname(os.path.abspath(__file__))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    newest = 0.0
    for f in sorted(glob.glob(os.path.join(src, "*_counter_collection.csv"))):
        newest = max(newest, os.path.getmtime(f))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc[k]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    summary = {}
    for k, v in pmc.items():
        d = {c: sum(x) / len(x) for c, x in v.items()}
        d["launches_seen"] = len(v["duration_us"])
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes_per_launch"] = 2.0 * d["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes_per_launch"] = d["WRITE_SIZE"] * 1024.0
        if "GRBM_GUI_ACTIVE" in d and d["duration_us"] > 0:
            d["clock_mhz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["duration_us"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                d["mfma_pipe_utilisation"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0)
        summary[k] = d
    if len(sys.argv) > 3:
        summary["_meta"] = {"workload_key": sys.argv[3], "command": "rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py",
                            "round": tag.split("_")[0], "head": git_head(os.path.dirname(out_dir)),
                            "collected_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(newest)) if newest else None}
    if summary:
        json.dump(summary, open(os.path.join(out_dir, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    for f in glob.glob(os.path.join(src, "*_kernel_stats.csv")):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w", newline="") as o:
            csv.writer(o).writerows(rows[:40])
    print("wrote", tag)


if __name__ == "__main__":
    main()
