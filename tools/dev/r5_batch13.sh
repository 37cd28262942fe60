#!/bin/bash
# round 5 batch 13: config 2 (8,16,1024,64) forward through the developer library's tuning points (one round of 512 workgroups at the shipped point)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5m}; mkdir -p $O
cd $R/tools
for rep in 1 2; do for v in 0 1 3 86 87 84 45; do
  echo -n "variant $v: "; timeout 120 ./fasn_harness bench 8 16 1024 1024 64 bf16 0 $v 400 2>&1 | tail -1
done; done 2>&1 | tee $O/c2_forward_tuning_points.log
