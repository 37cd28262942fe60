#!/bin/bash
# round 5 batch 7: fp32 bias at head dim 128 (4-wave forward, one-wave backward), config 4 with an fp32 ALiBi against the bf16 one,
# the XLNet surgery on the vector path, which limiter holds the clock (amd-smi throttle accumulators)
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5g}; mkdir -p $O
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
timeout 600 $H test 0 1 > $O/harness_test.log 2>&1; echo "harness test rc=$?"; grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test.log | head -20
{
for bk in 1 3; do echo "== config 4 (4,32,8192,128) bf16 n=0.5 ALiBi bias_kind $bk (1 = bf16, 3 = fp32) + key padding"; $H bench 4 32 8192 8192 128 1 0 0 30 1 0.5 4 $bk | tail -2; done
} 2>&1 | tee $O/c4_f32_bias.log
{
echo "=== MFMA-only launch"; bash $R/tools/limiter_probe.sh mfma
echo "=== M0 forward loop"; bash $R/tools/limiter_probe.sh 8 16 4096 4096 64 1 0 0 12000
echo "=== M0 backward loop"; bash $R/tools/limiter_probe.sh 8 16 4096 4096 64 1 0 0 10 1
} > $O/limiter.log 2>&1
cd $R && timeout 1500 python -m pytest tests/test_gpu_surgery.py tests/test_gpu_parity.py -m gpu -x -q -k "surgery or xlnet or bias or kernel_path" 2>&1 | tail -4 | tee $O/pytest.log
