#!/bin/bash
# usage: scripts/pmc_radon.sh <tag> <counters...>   (GPU box)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcr_$tag
rm -rf $out
timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/scripts/bench_radon.py 50 1024 768 2 > $GRAFT_REPO_ROOT/gpurun_out/pmcr_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections, re
rows=[]
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    m=re.search(r"(radon_kernel<[\w, ]+>)", r["Kernel_Name"])
    if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$GRAFT_REPO_ROOT/gpurun_out/pmcr_$tag.summary.txt","w") as f:
    for k,v in sorted(acc.items()):
        line = k + " " + ", ".join("%s=%.6g (n=%d)" % (c, sum(x)/len(x), len(x)) for c,x in sorted(v.items()))
        print(line); f.write(line+"\n")
PY
rm -rf $out
