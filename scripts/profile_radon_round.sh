#!/bin/bash
# usage (GPU box): scripts/profile_radon_round.sh <tag>  -- the two PMC passes of the Radon kernel (both arithmetic modes:
# scripts/bench_radon.py alternates them), summaries under gpurun_out/
set -e
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
scripts/pmc_radon.sh ${tag}_radon_sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
scripts/pmc_radon.sh ${tag}_radon_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
