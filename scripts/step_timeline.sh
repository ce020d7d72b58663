#!/bin/bash
# usage (GPU box): scripts/step_timeline.sh [env assignments...]  -- kernel start/end timestamps of the optimiser steps of
# bench.py (rocprofv3 --kernel-trace): median duration of every kernel of a step and of the gaps between them
# (BENCH_ARGS="--views 64 --size 512" as an env assignment: another workload)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in "$@"; do export "$a"; done
out=$R/gpurun_out/timeline
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-live-pmc $BENCH_ARGS > $R/gpurun_out/timeline.log 2>&1
python3 - <<PY
import csv, glob, collections, statistics as st
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("small_eval_kernel", "sum_pairs_split_kernel", "sum_pairs_kernel", "e1_kernel", "k01_kernel", "pairs_reference_kernel", "pairs_split_kernel", "pairs_kernel"):
        if k in n: return k
    return None
ours = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if short(r["Kernel_Name"])]
# a step ends with a sum kernel.  bench.py's timed blocks come first (record reuse as the library defaults), its
# reuse-off blocks near the end (5 + 3 x 20 timed steps, the event-timed pass, then 5 + 3 x 20 steps with reuse off):
# steps 10..60 are in the former, steps -70..-10 in the latter
ends_all = [i for i, o in enumerate(ours) if o[0].startswith("sum_pairs") or o[0] == "small_eval_kernel"]
for title, ends in (("library default (record reuse on)", ends_all[10:61]), ("ecc_metric_set_record_reuse(0)", ends_all[-70:-10])):
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in zip(ends, ends[1:]):
        step = ours[a + 1:b + 1]
        prev_end = ours[a][2]
        for nm, s, e in step:
            dur[nm].append((e - s) / 1e3)
            gap["start of %s after the previous step's sum" % nm].append((s - prev_end) / 1e3)
        gap["step (sum end -> sum end)"].append((ours[b][2] - prev_end) / 1e3)
        gap["sum start after the end of the step's last other kernel"].append((ours[b][1] - max([e for nm, s, e in step[:-1]] or [ours[b][1]])) / 1e3)
    print("== " + title)
    for k, v in dur.items(): print("kernel %-28s %8.1f us (median of %d)" % (k, st.median(v), len(v)))
    for k, v in gap.items(): print("       %-66s %8.1f us (median of %d)" % (k, st.median(v), len(v)))
PY
rm -rf $out
