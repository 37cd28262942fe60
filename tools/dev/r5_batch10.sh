#!/bin/bash
# round 5 batch 10: causal forward, final builds, alternating same-box A/B (variant 91 = 32 rows per wave, 92 = folded two-phase 64 rows per wave)
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5j}; mkdir -p $O
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
{
for rep in 1 2 3; do for v in 91 92; do
  for a in "64 16 4096 4096 64 1 1" "4 32 8192 8192 64 1 1" "2 16 16384 16384 64 1 1" "16 16 4096 4096 64 1 1" "8 16 4096 4096 64 0 1"; do $H bench $a $v 80 | tail -1; done
done; done
} 2>&1 | tee $O/causal_ab.log
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "256" 2>&1 | tail -3 | tee $O/pytest_256.log
