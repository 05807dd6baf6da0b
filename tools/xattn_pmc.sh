#!/bin/bash
# SQ / cache counter passes for the one-launch cross-attention kernel (separate rocprofv3 runs); usage: tools/xattn_pmc.sh <tag> [pre]; writes gpurun_out/pmc_<tag>_summary.txt
tag=$1
mode=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o p -- python3 $R/tools/_xattn_once.py 128 3 $mode > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* 2>&1 | grep -E "^#|xattn" > gpurun_out/pmc_${tag}_summary.txt
