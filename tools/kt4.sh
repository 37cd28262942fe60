#!/bin/bash
# per-kernel times of the D = 128 backward for plain / key padding / bias / both (same box): tools/kt4.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for f in "0 0" "1 0" "0 1" "1 1"; do
  echo "== mask bias = $f"; rm -rf /tmp/kt4
  rocprofv3 --kernel-trace --stats -d /tmp/kt4 -o kt -- $R/tools/fasn_harness bench 4 32 8192 8192 128 1 0 0 5 1 0.5 $f 0 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/kt4 fasn_ | sed 's/.*kernel void fasn:://' | cut -c1-150
done
