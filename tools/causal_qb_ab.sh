#!/bin/bash
# same-box A/B of the causal forward tuning points WITH paired blocks: variant 0 = shipped (32 rows per wave, 3 waves per SIMD),
# 85 = 64 rows per wave, 2 waves per SIMD (the plain kernel's tuning point; before the blocks were paired it lost to the tail)
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
for rep in 1 2; do for v in 0 85; do
  echo "== variant $v"
  $H bench 8 16 4096 4096 64 1 1 $v 200 | tail -1
  $H bench 8 16 4096 4096 64 0 1 $v 200 | tail -1
  $H bench 64 16 4096 4096 64 1 1 $v 30 | tail -1
  $H bench 4 32 8192 8192 64 1 1 $v 50 | tail -1
  $H bench 2 16 16384 16384 64 1 1 $v 30 | tail -1
done; done
