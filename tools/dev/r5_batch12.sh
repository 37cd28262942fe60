#!/bin/bash
# round 5 batch 12: the library with 16-byte epilogue stores through the whole GPU suite + harness self-test, then static wave priorities
# (tools/var/prio8: second half of the 8-wave forward workgroups; wsb / wsa: wave B / wave A of the two-wave backward kernels) at config 4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5l}; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
(cd tools && timeout 600 ./fasn_harness test 0 1 2>&1 | tail -4) | tee $O/harness_test.log
{
echo "=== c4 fwd: in-tree vs prio8"; bash tools/ab_libs.sh "bench.py --workload c4 --pass fwd --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes" . prio8
echo "=== c4 bwd: in-tree vs wsb vs wsa"; bash tools/ab_libs.sh "bench.py --workload c4 --pass bwd --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes" . wsb wsa
} 2>&1 | tee $O/prio_ab.log
