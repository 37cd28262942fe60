#!/bin/bash
# PMC counters of the one-pass backward (bwd_variant 0) next to the split kernels (4) at M0, dev library
R=${GRAFT_REPO_ROOT:-/root/repo}
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for bv in ${1:-0 4}; do
  echo "=== bwd_variant $bv"
  $R/tools/pmc_one.sh "$SQ1" 8 16 4096 4096 64 1 0 0 3 1 1.0 0 0 $bv | grep -v "fwd_kernel"
  $R/tools/pmc_one.sh "$SQ2" 8 16 4096 4096 64 1 0 0 3 1 1.0 0 0 $bv | grep -v "fwd_kernel"
done
