#!/bin/bash
# tools/mkvar.sh NAME "<extra -D flags>" file1.hip [file2.hip ...]: a variant libfasn.so under tools/var/NAME/ = the in-tree objects with
# the named sources recompiled with the extra flags (same-box A/B through tools/ab_libs.sh)
set -e
N=$1; F=$2; shift 2
C=/root/repo/flash-attention-softmax-n_amd/csrc; V=/root/repo/tools/var/$N; mkdir -p $V
OBJS=""
for o in $C/build/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do if [ "$(basename $f .hip)" = "$b" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I$C -Wno-unused-value -Wno-inline-asm $F -c $C/$b.hip -o $V/$b.o & use=$V/$b.o
  fi; done
  OBJS="$OBJS $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libfasn.so $OBJS
ls -la $V/libfasn.so
