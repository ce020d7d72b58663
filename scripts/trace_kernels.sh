#!/bin/bash
# usage (GPU box): scripts/trace_kernels.sh <script.py> [args...]  -- rocprofv3 --kernel-trace of a python script: per kernel name
# the median duration and the median gap to the end of the previous kernel on the device
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ktrace
rm -rf $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/ktrace.log 2>&1
python3 - <<PY
import csv, glob, collections, statistics as st, re
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev = None
for r in rows:
    m = re.search(r"(small_eval_kernel<[^>]*>|k01_kernel<\d+>|pairs_reference_kernel<[^>]*>|pairs_kernel<[^>]*>|sum_pairs\w*|e1_kernel|radon_kernel<[^>]*>|\w+_kernel\w*)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append((e - s) / 1e3)
    if prev is not None: gap[name].append((s - prev) / 1e3)
    prev = e
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s n=%5d  median %8.1f us  gap before (median) %8.1f us" % (k, len(v), st.median(v), st.median(gap[k]) if gap[k] else 0))
PY
rm -rf $out
