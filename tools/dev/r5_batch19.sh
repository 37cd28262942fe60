#!/bin/bash
# round 5 batch 19: causal forward, folded two-phase walk with eight waves per workgroup (developer variant 95) against the shipped rule (0) at C3 / C5 and longer
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5s}; mkdir -p $O
cd $R/tools
timeout 300 ./fasn_harness test 95 1 2>&1 | tail -2 | tee $O/causal_fold_8_waves.log
for rep in 1 2; do for shape in "8 16 4096 4096 64 0" "64 16 4096 4096 64 1" "4 32 8192 8192 64 1" "2 16 16384 16384 64 1"; do for v in 0 95; do
  set -- $shape
  echo -n "variant $v: "; timeout 120 ./fasn_harness bench $1 $2 $3 $4 $5 $6 1 $v 300 2>&1 | tail -1
done; done; done 2>&1 | tee -a $O/causal_fold_8_waves.log
