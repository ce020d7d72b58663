#!/bin/bash
# usage (GPU box): scripts/profile_round.sh <tag>   -- kernel stats + PMC passes of bench.py, summaries under gpurun_out/
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/stats_$tag
rm -rf $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-live-pmc > $R/gpurun_out/stats_$tag.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -1 "$f" > $R/gpurun_out/${tag}_kernel_stats.csv
grep -E "pairs_kernel|k01_kernel|radon_kernel|sum_pairs|e1_kernel|dtr_border|preprocess_kernel|ramp_kernel" "$f" >> $R/gpurun_out/${tag}_kernel_stats.csv || true
cat $R/gpurun_out/${tag}_kernel_stats.csv
rm -rf $out
cd $R
scripts/pmc_pass.sh ${tag}_fetch FETCH_SIZE
scripts/pmc_pass.sh ${tag}_write WRITE_SIZE
scripts/pmc_pass.sh ${tag}_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
scripts/pmc_pass.sh ${tag}_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
scripts/pmc_pass.sh ${tag}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
scripts/pmc_pass.sh ${tag}_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD
