#!/bin/bash
# one PMC pass of one harness bench line: tools/pmc_one.sh "<counters>" <harness bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; C="$1"; shift; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc1
rocprofv3 --pmc $C -d /tmp/pmc1 -o pmc -- $R/tools/fasn_harness bench "$@" > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc1 fasn_ | grep -v "calls=" | sed 's/.*\] //' | cut -c1-120
