#!/bin/bash
# round 5 batch 21: delta computed by the one-wave dQ kernels of head dims <= 64 (in-tree) against the separate delta launch (tools/var/nofuse): whole
# GPU suite first, then BERT-like shapes with key-padding masks by graph replay
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5u}; mkdir -p $O
cd $R; P=flash-attention-softmax-n_amd/libfasn.so
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu_tail.log
cp $P /tmp/intree.so
{
for rep in 1 2; do for d in . nofuse; do
  if [ "$d" = "." ]; then cp /tmp/intree.so $P; else cp tools/var/$d/libfasn.so $P; fi
  echo "== library $d"; python tools/bench_small_shapes.py 2>&1 | grep ms_per_step
done; done
} 2>&1 | tee $O/delta_in_one_wave_dq_ab.log
cp /tmp/intree.so $P
