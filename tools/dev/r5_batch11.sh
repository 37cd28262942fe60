#!/bin/bash
# round 5 batch 11: 16-byte output stores (tools/var/wideo: FASN_WIDE_O=1 in the D = 64 / 128 forward and the pipelined D = 64 backward) against the in-tree library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5k}; mkdir -p $O
cd $R
for wp in "m0 fwd" "c2 fwd" "c3 fwd" "c5 fwd" "m0 bwd" "c2 bwd" "c4 fwd"; do
  set -- $wp
  echo "=== $1 $2"; bash tools/ab_libs.sh "bench.py --workload $1 --pass $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes" . wideo . wideo
done 2>&1 | tee $O/wide_stores_ab.log
cp tools/var/wideo/libfasn.so /tmp/wideo.so; cp flash-attention-softmax-n_amd/libfasn.so /tmp/intree_keep.so; cp /tmp/wideo.so flash-attention-softmax-n_amd/libfasn.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or analytic or oracle or causal or properties or backward" 2>&1 | tail -3 | tee $O/pytest_wideo.log
cp /tmp/intree_keep.so flash-attention-softmax-n_amd/libfasn.so
