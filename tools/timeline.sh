#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-tl}; O=$R/gpurun_out/$T; mkdir -p $O
H=$R/tools/fasn_harness
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
shift
{
while [ $# -gt 0 ]; do timeout 600 $H timeline $1; shift; done
} > $O/log.txt 2>&1
cat $O/log.txt
