#!/bin/bash
# usage (GPU box): scripts/pmc_memory_path.sh  -- PMC passes on the pair kernel's memory path, get_ij order and tile order
cd $GRAFT_REPO_ROOT
for t in 0 32; do
  export ECC_PAIR_TILE=$t
  scripts/pmc_pass.sh mp${t}_utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum | grep pairs_kernel
  scripts/pmc_pass.sh mp${t}_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum | grep pairs_kernel
  scripts/pmc_pass.sh mp${t}_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum | grep pairs_kernel
  scripts/pmc_pass.sh mp${t}_tcc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum | grep pairs_kernel
done
export ECC_PAIR_TILE=0
scripts/pmc_pass.sh mp0_ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum | grep pairs_kernel
scripts/pmc_pass.sh mp0_tcpstall TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum | grep pairs_kernel
scripts/pmc_pass.sh mp0_busy GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCC_BUSY_sum | grep pairs_kernel
