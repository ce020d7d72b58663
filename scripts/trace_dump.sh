#!/bin/bash
# usage (GPU box): scripts/trace_dump.sh <first> <count> <script.py> [args...] -- rocprofv3 --kernel-trace of a python script; prints
# `count` consecutive kernel records (start offset, duration, gap to the previous end, grid, name) from record `first` on
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ktrace
rm -rf $out
first=$1; count=$2; shift; shift
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/ktrace.log 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if re.search(r"pairs_kernel|pairs_split_kernel|k01_kernel|sum_pairs|e1_kernel|small_eval", r["Kernel_Name"])]
sel = ours[$first:$first + $count]
t0 = int(sel[0]["Start_Timestamp"]); prev = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(small_eval_kernel|k01_kernel<\d+>|pairs_reference_kernel|pairs_split_kernel|pairs_kernel|sum_pairs\w*|e1_kernel)", r["Kernel_Name"])
    g = int(float(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))
    print("%9.1f us  dur %7.1f  gap %7.1f  grid %8d  queue %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, g, r.get("Queue_Id", "?"), m.group(1)))
    prev = e
PY
rm -rf $out
