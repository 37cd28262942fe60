#!/bin/bash
# same-box A/B of a python benchmark: tools/old/libfasn.so (previous build) vs the in-tree library
# usage (via gpurun): tools/ab_py.sh tools/bench_train_step.py
P=flash-attention-softmax-n_amd/libfasn.so
cp $P /tmp/new.so
for lib in old new old new; do
  if [ $lib = old ]; then cp tools/old/libfasn.so $P; else cp /tmp/new.so $P; fi
  echo "== $lib"; python "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/new.so $P
