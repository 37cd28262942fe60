#!/bin/bash
# one PMC pass of a bench.py line for each variant library: tools/pmc_py.sh "<counters>" "<bench.py args>" var1 var2 ... ("." = in-tree)
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/flash-attention-softmax-n_amd/libfasn.so; C="$1"; A="$2"; shift 2
cp $P /tmp/intree.so; cd /tmp; export TMPDIR=/tmp
for d in "$@"; do
  if [ "$d" = "." ]; then cp /tmp/intree.so $P; else cp $R/tools/var/$d/libfasn.so $P; fi
  rm -rf /tmp/pmcpy; rocprofv3 --pmc $C -d /tmp/pmcpy -o pmc -- python $R/bench.py $A --steps 3 --warmup 1 --no-cpu-baseline --no-extra-passes > /dev/null 2>&1
  echo "== $d"; python3 $R/tools/pmc_summary.py /tmp/pmcpy fasn_ | sed 's/.*\] //' | cut -c1-200
done
cp /tmp/intree.so $P
