#!/bin/bash
# per-kernel times of one harness bench line: tools/kt_one.sh <harness bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt1
rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt -- $R/tools/fasn_harness bench "$@" > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/kt1 fasn_ | sed 's/.*kernel void fasn:://' | cut -c1-150
