#!/bin/bash
# round 5 batch 23: progress-based wave priority (developer variants 96 / 97) for one-round forward launches: config 2 and neighbours, timeline + timing
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5y2}; mkdir -p $O
cd $R/tools
{
timeout 300 ./fasn_harness test 96 1 2>&1 | tail -1
for v in 0 96; do ./fasn_harness timeline 8 16 1024 1024 64 1 0 $v 2>&1 | tail -5; done
for rep in 1 2 3; do for shape in "8 16 1024 1024" "4 16 2048 2048" "16 16 512 512" "8 16 4096 4096" "2 16 4096 4096"; do for v in 0 96; do
  set -- $shape
  echo -n "variant $v: "; timeout 120 ./fasn_harness bench $1 $2 $3 $4 64 1 0 $v 1000 2>&1 | tail -1
done; done; done
} 2>&1 | tee $O/progress_priority_one_round.log
