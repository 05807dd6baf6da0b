#!/bin/bash
# PMC passes (separate rocprofv3 runs: SQ has 8 slots, TCC 4) for one program; usage: tools/pmc_run.sh <tag> <python script> [args...]
# Writes gpurun_out/pmc_<tag>_summary.txt.  Run on the GPU box from the repo root.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
script=$1; shift; case "$script" in /*) ;; *) script=$R/$script;; esac
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o p -- python3 "$script" "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* > gpurun_out/pmc_${tag}_summary.txt 2>&1
