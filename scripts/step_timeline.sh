#!/bin/bash
# usage (GPU box): scripts/step_timeline.sh [env assignments...]  -- kernel start/end timestamps of the optimiser steps of
# bench.py (rocprofv3 --kernel-trace): median duration of every kernel of a step and of the gaps between them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in "$@"; do export "$a"; done
out=$R/gpurun_out/timeline
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-live-pmc > $R/gpurun_out/timeline.log 2>&1
python3 - <<PY
import csv, glob, collections, statistics as st
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("e1_kernel", "k01_kernel", "pairs_kernel", "pairs_reference_kernel", "sum_pairs_split_kernel", "sum_pairs_kernel"):
        if k in n: return k
    return None
ours = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if short(r["Kernel_Name"])]
# a step ends with a sum kernel; take the last 40 steps
ends = [i for i, o in enumerate(ours) if o[0].startswith("sum_pairs")]
ends = ends[-41:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(ends, ends[1:]):
    step = ours[a + 1:b + 1]
    for k, (nm, s, e) in enumerate(step):
        dur[nm].append((e - s) / 1e3)
        prev = ours[a + k]
        gap[prev[0] + " -> " + nm].append((s - prev[2]) / 1e3)
    gap["step (sum end -> sum end)"].append((ours[b][2] - ours[a][2]) / 1e3)
for k, v in dur.items(): print("kernel %-28s %8.1f us (median of %d)" % (k, st.median(v), len(v)))
for k, v in gap.items(): print("gap    %-48s %8.1f us (median of %d)" % (k, st.median(v), len(v)))
PY
rm -rf $out
