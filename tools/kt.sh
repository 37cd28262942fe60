#!/bin/bash
# usage (via gpurun): tools/kt.sh TAG "<bench.py args>"   -> per-kernel average durations (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py $2 --no-cpu-baseline --no-extra-passes > $O/kt.log 2>&1
python3 $R/tools/pmc_summary.py $O/kt fasn_ > $O/kt_summary.txt 2>&1
cat $O/kt_summary.txt
