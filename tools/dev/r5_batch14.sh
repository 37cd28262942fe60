#!/bin/bash
# round 5 batch 14: fp32 bias at D = 128 on the 8-wave register-staged forward (in-tree) against the 4-wave first build (tools/var/bf32w4); parity of the fp32-bias tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5n}; mkdir -p $O
cd $R; P=flash-attention-softmax-n_amd/libfasn.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surgery.py -m gpu -x -q -k "f32 or fp32 or bias or surgery or xlnet or kernel_path" 2>&1 | tail -3 | tee $O/pytest_bias.log
cp $P /tmp/intree.so
{
for rep in 1 2; do for d in . bf32w4; do
  if [ "$d" = "." ]; then cp /tmp/intree.so $P; else cp tools/var/$d/libfasn.so $P; fi
  echo "== library $d"; python tools/bench_fp32_bias.py both d128
done; done
} 2>&1 | tee $O/fp32_bias_8wave_ab.log
cp /tmp/intree.so $P
