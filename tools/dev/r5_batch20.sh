#!/bin/bash
# round 5 batch 20: where eight waves per workgroup (developer variant 93) overtake four (0) in the plain D = 64 forward: launches of 4 .. 32 rounds
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5t}; mkdir -p $O
cd $R/tools
for rep in 1 2; do for shape in "8 16 4096 4096" "12 16 4096 4096" "16 16 4096 4096" "32 16 4096 4096" "8 16 8192 8192" "4 16 16384 16384" "16 16 2048 2048" "64 16 1024 1024"; do for v in 0 93; do
  set -- $shape
  echo -n "variant $v: "; timeout 120 ./fasn_harness bench $1 $2 $3 $4 64 1 0 $v 400 2>&1 | tail -1
done; done; done 2>&1 | tee $O/plain_forward_8_waves_crossover.log
