#!/bin/bash
# SQ counters of the forward at M0 / C3 / C5 (harness, dev library = same kernels as libfasn.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for cfg in "8 16 4096 4096 64 1 0" "8 16 4096 4096 64 0 1" "64 16 4096 4096 64 1 1"; do
  echo "=== fwd $cfg"
  $R/tools/pmc_one.sh "$SQ1" $cfg 0 5 0
  $R/tools/pmc_one.sh "$SQ2" $cfg 0 5 0
  $R/tools/fasn_harness bench $cfg 0 50 0 | tail -1
done
