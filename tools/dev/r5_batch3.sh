#!/bin/bash
# round 5 batch 3: causal forward with the folded two-phase walk (variant 92) against 32 rows per wave (91) and the unfolded 64 rows per wave (90)
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5c}; mkdir -p $O
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
timeout 600 $H test 92 1 > $O/harness_test_v92.log 2>&1; echo "harness test variant 92 rc=$?"; grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test_v92.log | head -20
timeout 600 $H test 0 1 > $O/harness_test_v0.log 2>&1; echo "harness test variant 0 rc=$?"; grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test_v0.log | head -20
bash $R/tools/r5_causal.sh ${1:-r5c} "91 92 90"
{
for v in 91 92; do for a in "4 32 8192 8192 64 1 1" "2 16 16384 16384 64 1 1" "8 16 2048 2048 64 1 1" "16 16 1024 1024 64 1 1" "8 16 4096 4096 64 1 1"; do $H bench $a $v 100 | tail -1; done; done
} 2>&1 | tee $O/causal_more.log
cd $R && timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "causal or golden or analytic or properties" 2>&1 | tail -5 | tee $O/pytest_causal.log
