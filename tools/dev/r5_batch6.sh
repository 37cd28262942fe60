#!/bin/bash
# round 5 batch 6: after the register-staged K/V of a rotated second walk was fixed (head dims <= 64) - harness parity, the whole GPU test suite
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5f}; mkdir -p $O
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
timeout 600 $H test 0 1 > $O/harness_test.log 2>&1; echo "harness test rc=$?"; grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test.log | head -20
FASN_KPROT=0 timeout 600 $H test 0 1 > $O/harness_test_kprot0.log 2>&1; echo "harness test (no rotation) rc=$?"; grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test_kprot0.log | head -5
cd $R && timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
