#!/bin/bash
# same-box A/B of variant libraries through a python benchmark: tools/ab_libs.sh "<python args>" var1 var2 ...  ("." = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/flash-attention-softmax-n_amd/libfasn.so
cp $P /tmp/intree.so
A="$1"; shift
for rep in 1 2; do for d in "$@"; do
  if [ "$d" = "." ]; then cp /tmp/intree.so $P; else cp $R/tools/var/$d/libfasn.so $P; fi
  echo -n "$d: "; python $A 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo
done; done
cp /tmp/intree.so $P
