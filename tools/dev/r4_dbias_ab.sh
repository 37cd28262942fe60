#!/bin/bash
# same-box timing of variant libraries (tools/mkvar.sh) through tools/bench_dbias.py: tools/r4_dbias_ab.sh var1 var2 ...  ("." = in-tree)
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/flash-attention-softmax-n_amd/libfasn.so
cp $P /tmp/intree.so
for d in "$@"; do
  if [ "$d" = "." ]; then cp /tmp/intree.so $P; else cp $R/tools/var/$d/libfasn.so $P; fi
  echo "== $d"; timeout 120 python $R/tools/bench_dbias.py 2>&1 | grep "dbias kernel" | sed 's/forward + backward, ALiBi .H,L,S. bias + key padding: //'
done
cp /tmp/intree.so $P
