#!/bin/bash
# usage (GPU box): scripts/rccl_first_contact.sh <tag>
# The collective path on ONE GPU: the one-rank RCCL test, bench.py --force-collective next to the plain N = 1 line on the
# same box, and a rocprofv3 kernel trace of a short --force-collective run (pair kernel -> RCCL kernel -> publish_scalar_kernel
# in stream order).  Everything lands under gpurun_out/.
set -e
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_rccl_one_rank.py -x -q -m gpu > gpurun_out/${tag}_rccl_test.log 2>&1 || { tail -40 gpurun_out/${tag}_rccl_test.log; exit 1; }
tail -3 gpurun_out/${tag}_rccl_test.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > gpurun_out/${tag}_bench_plain_same_box.json 2> gpurun_out/${tag}_bench_plain_same_box.err
python bench.py --force-collective --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > gpurun_out/${tag}_bench_1rank_rccl.json 2> gpurun_out/${tag}_bench_1rank_rccl.err
python - <<PY
import json
last = lambda f: json.loads([t for t in open(f) if t.startswith("{")][-1])
a = last("gpurun_out/${tag}_bench_plain_same_box.json"); b = last("gpurun_out/${tag}_bench_1rank_rccl.json")
print("plain N=1: %.1f /s   one-rank RCCL: %.1f /s (%s, ranks seen %s)   ratio %.4f   other exchange: %s" % (
    a["value"], b["value"], b["config"]["sum_exchange"], b["config"]["ranks_seen_by_collective_backend"], b["value"] / a["value"],
    b["timing"].get("other_exchange")))
PY
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/trace_${tag}_rccl
rm -rf $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --force-collective --exchange collective --steps 4 --warmup 2 --blocks 2 --no-cpu-baseline --no-live-pmc --no-power > $R/gpurun_out/${tag}_rccl_trace.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 60 kernels of the run: the timed steps (pair kernel -> collective -> publish) and the block-time reductions
t0 = int(rows[0]["Start_Timestamp"])
with open("$R/gpurun_out/${tag}_rccl_kernel_order.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace of bench.py --force-collective --exchange collective --steps 4 --warmup 2 --blocks 2\n")
    f.write("# start_us  dur_us  stream/queue  kernel   (last 80 dispatches)\n")
    for r in rows[-80:]:
        f.write("%12.1f %9.1f  q%s  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                             r.get("Queue_Id", "?"), r["Kernel_Name"][:110]))
names = {}
for r in rows:
    names[r["Kernel_Name"][:80]] = names.get(r["Kernel_Name"][:80], 0) + 1
print({k: v for k, v in names.items() if "ccl" in k.lower() or "publish" in k or "pairs_kernel" in k})
PY
tail -30 $R/gpurun_out/${tag}_rccl_kernel_order.txt
rm -rf $out
# RCCL device code in the stream: the same sequence with the reduction op AVG (see scripts/rccl_one_rank_avg_trace.py)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/scripts/rccl_one_rank_avg_trace.py > $R/gpurun_out/${tag}_rccl_avg_trace.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open("$R/gpurun_out/${tag}_rccl_avg_kernel_order.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace of scripts/rccl_one_rank_avg_trace.py (one-rank RCCL group, all_reduce(AVG)); last 40 dispatches\n")
    f.write("# start_us  dur_us  queue  kernel\n")
    for r in rows[-40:]:
        f.write("%12.1f %9.1f  q%s  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                             r.get("Queue_Id", "?"), r["Kernel_Name"][:120]))
PY
grep -a "one-rank RCCL" $R/gpurun_out/${tag}_rccl_avg_trace.log
tail -16 $R/gpurun_out/${tag}_rccl_avg_kernel_order.txt
rm -rf $out
