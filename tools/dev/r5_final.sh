#!/bin/bash
# end of round 5: the whole GPU suite, smoke, the harness self-test (developer library), then the profile round of the same library
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r5z}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu_tail.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
(cd tools && timeout 600 ./fasn_harness test 0 1 2>&1 | tail -3) | tee $O/harness_test.log
bash tools/prof_round4.sh $T all
