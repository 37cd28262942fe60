#!/bin/bash
# round 5 batch 17: delta computed in the prologue of the pipelined dQ kernel (in-tree: no delta launch on the D = 64 plain / causal path) against the
# separate delta kernel (tools/var/deltak); parity first (whole GPU suite)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5q}; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu_tail.log
for wp in "c2 bwd" "m0 bwd" "c3 bwd" "c5 bwd"; do
  set -- $wp
  echo "=== $1 $2"; bash tools/ab_libs.sh "bench.py --workload $1 --pass $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes" . deltak . deltak
done 2>&1 | tee $O/delta_in_dq_ab.log
