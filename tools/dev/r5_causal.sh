#!/bin/bash
# round 5: causal forward at the plain kernel's tuning point (64 rows per wave, 2 workgroups per CU: variant 90) against the shipped one
# (32 rows per wave, 3 per CU: variant 91 = variant 0), with per-workgroup timelines.   usage: tools/r5_causal.sh TAG [variants]
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5causal}; mkdir -p $O
V=${2:-"91 90"}
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
{
for rep in 1 2; do for v in $V; do
  for a in "8 16 4096 4096 64 0 1" "64 16 4096 4096 64 1 1" "8 16 4096 4096 64 1 0" "8 16 1024 1024 64 1 0"; do $H bench $a $v 100 | tail -1; done
done; done
for v in $V; do
  for a in "8 16 4096 4096 64 1 1" "64 16 4096 4096 64 1 1"; do echo "-- timeline $a variant $v"; $H timeline $a $v | head -5 | cut -c1-260; done
done
} 2>&1 | tee $O/causal.log
