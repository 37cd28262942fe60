#!/bin/bash
# same-box A/B of variant libraries: tools/ab_dirs.sh "<harness bench args>" dir1 dir2 ...   ("." = the in-tree library)
A="$1"; shift
for rep in 1 2; do for d in "$@"; do
  if [ "$d" = "." ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$PWD/tools/$d; fi
  echo -n "$d: "; tools/fasn_harness bench $A 2>&1 | grep -E "bwd|fwd" | tr '\n' ' '; echo
done; done
