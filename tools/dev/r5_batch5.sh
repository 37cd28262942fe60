#!/bin/bash
# round 5 batch 5: fp32 bias next to 16-bit q / k / v on the vector path (head dims <= 64): harness parity (forward + backward), the package's
# bias tests, and its cost against the same bias in the q dtype
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5e}; mkdir -p $O
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
timeout 600 $H test 0 1 > $O/harness_test.log 2>&1; echo "harness test rc=$?"; grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test.log | head -20
{
for bk in 1 3; do echo "== (4,16,4096,64) bf16 n=0.5, ALiBi [H,L,S] bias_kind $bk (1 = bf16, 3 = fp32), key padding"; $H bench 4 16 4096 4096 64 1 0 0 50 1 0.5 4 $bk | tail -2; done
for bk in 1 3; do echo "== (4,16,4096,64) bf16 n=1, ALiBi bias_kind $bk, no mask"; $H bench 4 16 4096 4096 64 1 0 0 50 1 1.0 0 $bk | tail -2; done
} 2>&1 | tee $O/f32_bias_cost.log
cd $R && timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surgery.py -m gpu -x -q -k "bias or mask or surgery or kernel_path or fuzz or random" 2>&1 | tail -6 | tee $O/pytest_bias.log
