#!/bin/bash
# usage (GPU box): scripts/pmc_any.sh <tag> "<counters...>" <script.py> [args...]
# One rocprofv3 --kernel-trace --pmc pass over a python script; per kernel name (first 90 characters) the per-dispatch
# average of every counter and of the duration -> gpurun_out/pmc_<tag>.summary.txt
set -e
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
rm -rf $out
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
per=collections.defaultdict(float)
for r in rows:
    per[(r["Kernel_Name"][:90] + (" grid=%s" % r["Grid_Size"] if "Grid_Size" in r else ""), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for (name, d, c), v in per.items(): acc[name][c].append(v)
dur=collections.defaultdict(list)
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        g = [int(float(r.get(k) or 1)) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")] if "Grid_Size_X" in r else None
        dur[r["Kernel_Name"][:90] + (" grid=%d" % (g[0] * g[1] * g[2]) if g else "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$R/gpurun_out/pmc_$tag.summary.txt","w") as f:
    for k,v in sorted(acc.items()):
        line = k + " | " + ", ".join("%s=%.6g (n=%d)" % (c, sum(x)/len(x), len(x)) for c,x in sorted(v.items()))
        if dur.get(k): line += " | avg %.2f us" % (sum(dur[k]) / len(dur[k]) / 1e3)
        print(line); f.write(line+"\n")
PY
rm -rf $out
