#!/bin/bash
# round 5 batch 16: config 2 forward, the developer library's 8-wave tuning points (13 15 16 17 18 19) against the shipped rule (0)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5p}; mkdir -p $O
cd $R/tools
for rep in 1 2; do for v in 0 13 15 16 17 18 19 40 43 2; do
  echo -n "variant $v: "; timeout 120 ./fasn_harness bench 8 16 1024 1024 64 1 0 $v 400 2>&1 | tail -1
done; done 2>&1 | tee $O/c2_forward_tuning_points_8wave.log
