#!/bin/bash
# usage (GPU box): scripts/profile_round.sh <tag>   -- kernel stats + PMC passes of bench.py, summaries under gpurun_out/
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/stats_$tag
rm -rf $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-live-pmc --no-power > $R/gpurun_out/stats_$tag.log 2>&1
# per-kernel statistics from the kernel trace, kernels of this library only; a name launched with clearly different grids
# (pairs_kernel: all pairs / the list of the moved view's pairs; radon_kernel: sub-batches) gets one line per grid
python3 - <<PY
import csv, glob, collections, re, statistics as st
rows = []
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    m = re.search(r"(pairs_split_kernel<[\w, ]+>|pairs_kernel<[\w, ]+>|pairs_reference_kernel<[\w, ]+>|pairs_reference_wide_kernel<[\w, ]+>|small_eval_kernel<[\w, ]+>|k01_kernel<\d+>|radon_kernel<[\w, ]+>|sum_pairs\w*kernel|e1_kernel|dtr_border_kernel|preprocess\w*kernel(?:<[-\w, ]+>)?|ramp_kernel<[\w, ]+>|direct_\w+kernel)", r["Kernel_Name"])
    if not m: continue
    g = [int(float(r.get(k) or 1)) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")] if "Grid_Size_X" in r else [int(float(r.get("Grid_Size") or 0)), 1, 1]
    acc[(m.group(1), g[0] * g[1] * g[2])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$R/gpurun_out/${tag}_kernel_stats.csv", "w") as f:
    f.write('"Name","GridThreads","Calls","TotalDurationNs","AverageNs","MedianNs","MinNs","MaxNs"\n')
    for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%d,%d,%.1f,%.1f,%d,%d\n' % (name, grid, len(v), sum(v), sum(v) / len(v), st.median(v), min(v), max(v)))
PY
cat $R/gpurun_out/${tag}_kernel_stats.csv
# rocprofv3's own --stats table (aggregated by name only), our kernels
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -1 "$f" > $R/gpurun_out/${tag}_kernel_stats_rocprofv3.csv
grep -E "small_eval_kernel|pairs_kernel|pairs_split_kernel|pairs_reference_kernel|pairs_reference_wide_kernel|k01_kernel|radon_kernel|sum_pairs|e1_kernel|dtr_border|preprocess_kernel|ramp_kernel" "$f" >> $R/gpurun_out/${tag}_kernel_stats_rocprofv3.csv || true
rm -rf $out
cd $R
scripts/pmc_pass.sh ${tag}_fetch FETCH_SIZE
scripts/pmc_pass.sh ${tag}_write WRITE_SIZE
scripts/pmc_pass.sh ${tag}_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
scripts/pmc_pass.sh ${tag}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
scripts/pmc_pass.sh ${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
scripts/pmc_pass.sh ${tag}_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD
