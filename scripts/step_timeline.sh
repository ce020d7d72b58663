#!/bin/bash
# usage (GPU box): scripts/step_timeline.sh  -- kernel start/end timestamps of a few optimiser steps (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/timeline
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/timeline.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# find the timed steps: sequences e1 -> k01 -> pairs -> sum
idx = [i for i, n in enumerate(names) if "e1_kernel" in n]
steps = []
for i in idx:
    if i + 3 < len(rows) and "k01_kernel" in names[i+1] and "pairs_kernel" in names[i+2] and "sum_pairs" in names[i+3]:
        steps.append(i)
steps = steps[8:20]
import statistics as st
def t(r, k): return int(r[k])
acc = {"e1": [], "gap e1->k01": [], "k01": [], "gap k01->pairs": [], "pairs": [], "gap pairs->sum": [], "sum": [], "sum end -> next e1": []}
for a, i in enumerate(steps):
    e1, k01, pr, sm = rows[i], rows[i+1], rows[i+2], rows[i+3]
    acc["e1"].append(t(e1,"End_Timestamp")-t(e1,"Start_Timestamp"))
    acc["gap e1->k01"].append(t(k01,"Start_Timestamp")-t(e1,"End_Timestamp"))
    acc["k01"].append(t(k01,"End_Timestamp")-t(k01,"Start_Timestamp"))
    acc["gap k01->pairs"].append(t(pr,"Start_Timestamp")-t(k01,"End_Timestamp"))
    acc["pairs"].append(t(pr,"End_Timestamp")-t(pr,"Start_Timestamp"))
    acc["gap pairs->sum"].append(t(sm,"Start_Timestamp")-t(pr,"End_Timestamp"))
    acc["sum"].append(t(sm,"End_Timestamp")-t(sm,"Start_Timestamp"))
    if a + 1 < len(steps):
        acc["sum end -> next e1"].append(t(rows[steps[a+1]],"Start_Timestamp")-t(sm,"End_Timestamp"))
for k, v in acc.items():
    print("%-22s %8.1f us (median of %d)" % (k, st.median(v)/1e3, len(v)))
PY
rm -rf $out
