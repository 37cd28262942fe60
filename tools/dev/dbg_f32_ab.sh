#!/bin/bash
# debug A/B: the fp32 general-mode check with the in-tree library and with variant libraries tools/var/<name>/libfasn.so
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/flash-attention-softmax-n_amd/libfasn.so
cp $P /tmp/intree.so
echo "== in-tree"; python $R/tools/dev/dbg_f32.py 2>&1 | grep "D=64 (2" | cut -c1-120
for v in "$@"; do
cp $R/tools/var/$v/libfasn.so $P
echo "== $v"; python $R/tools/dev/dbg_f32.py 2>&1 | grep "D=64 (2\|Error" | cut -c1-120
done
cp /tmp/intree.so $P
