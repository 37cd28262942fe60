#!/bin/bash
# round 5 batch 9: full GPU suite after the per-key bias gradient change; causal pairing A/B at C3 for the folded kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5i}; mkdir -p $O
cd $R && timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.log
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
{
for pm in 1 0; do for v in 91 92; do
  echo "== FASN_PAIR=$pm variant $v"
  FASN_PAIR=$pm $H bench 8 16 4096 4096 64 0 1 $v 200 | tail -1
  FASN_PAIR=$pm $H bench 8 16 4096 4096 64 1 1 $v 200 | tail -1
  FASN_PAIR=$pm $H bench 64 16 4096 4096 64 1 1 $v 30 | tail -1
done; done
} 2>&1 | tee $O/causal_pairing.log
