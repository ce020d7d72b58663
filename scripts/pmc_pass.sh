#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counters...>   (GPU box; one rocprofv3 --pmc pass, kernel-trace only)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --no-power > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections, re
rows=[]
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
def grid(r):
    return int(float(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))
# pairs_kernel is launched twice per step since round 3 (all pairs, and the list of the moved view's pairs on the side
# stream): per name only the dispatches of the larger kind count (grid at least half the largest)
gmax=collections.defaultdict(int)
for r in rows: gmax[r["Kernel_Name"]]=max(gmax[r["Kernel_Name"]], grid(r))
per=collections.defaultdict(float)   # a dispatch's counter can come in several rows (one per XCC group): they add up
for r in rows:
    k=r["Kernel_Name"]
    if 2*grid(r) < gmax[k]: continue
    m=re.search(r"(pairs_split_kernel<[\w, ]+>|pairs_kernel<[\w, ]+>|small_eval_kernel<[\w, ]+>|k01_kernel<\d+>|k01_kernel|radon_kernel<[\w, ]+>|sum_pairs\w*kernel|e1_kernel|dtr_border_kernel|preprocess_kernel<[-\w, ]+>)", k)
    if m:
        per[(m.group(1), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for (name, d, c), v in per.items(): acc[name][c].append(v)
with open("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.summary.txt","w") as f:
    for k,v in sorted(acc.items()):
        line = k + " " + ", ".join("%s=%.6g (n=%d)" % (c, sum(x)/len(x), len(x)) for c,x in sorted(v.items()))
        print(line); f.write(line+"\n")
PY
rm -rf $out
