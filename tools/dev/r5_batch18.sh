#!/bin/bash
# round 5 batch 18: the M0 forward with eight waves per workgroup (developer variants 93 / 94) against the shipped four (0), causal too
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5r}; mkdir -p $O
cd $R/tools
for rep in 1 2 3; do for v in 0 93 94; do
  echo -n "variant $v: "; timeout 120 ./fasn_harness bench 8 16 4096 4096 64 1 0 $v 2000 2>&1 | tail -1
done; done 2>&1 | tee $O/m0_forward_8_waves.log
for v in 0 93; do echo -n "variant $v: "; timeout 120 ./fasn_harness bench 64 16 4096 4096 64 1 0 $v 300 2>&1 | tail -1; done 2>&1 | tee -a $O/m0_forward_8_waves.log
timeout 300 ./fasn_harness test 93 1 2>&1 | tail -2 | tee -a $O/m0_forward_8_waves.log
