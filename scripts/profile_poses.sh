#!/bin/bash
# usage (GPU box): scripts/profile_poses.sh <tag>  -- kernel trace of scripts/pose_batch_breakdown.py (the batched pose path,
# csrc/ecc_poses.hip), per kernel AND grid, summary under gpurun_out/<tag>_pose_kernel_stats.csv
set -e
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pose_stats_$tag
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/scripts/pose_batch_breakdown.py > $R/gpurun_out/${tag}_pose_breakdown_profiled.txt 2>&1
python3 - <<PY
import csv, glob, collections, re, statistics as st
rows = []
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    g = [int(float(r.get(k) or 1)) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")] if "Grid_Size_X" in r else [int(float(r.get("Grid_Size") or 0)), 1, 1]
    acc[(name, g[0] * g[1] * g[2])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$R/gpurun_out/${tag}_pose_kernel_stats.csv", "w") as f:
    f.write('"Name","GridThreads","Calls","TotalDurationNs","AverageNs","MedianNs","MinNs","MaxNs"\n')
    for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%d,%d,%.1f,%.1f,%d,%d\n' % (name, grid, len(v), sum(v), sum(v) / len(v), st.median(v), min(v), max(v)))
PY
cat $R/gpurun_out/${tag}_pose_kernel_stats.csv
rm -rf $out
