#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_fwd.sh OUTDIR "<harness bench args>"
# kernel trace + separate PMC passes (never combined with tracing domains other than kernel dispatch).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
H="$R/tools/fasn_harness bench $2"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $H > $O/kt.log 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d" " -f1); rocprofv3 --pmc $set -d $O/pmc_$n -o pmc -- $H > $O/pmc_$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $O fasn_ > $O/summary.txt 2>&1
cat $O/summary.txt
