#!/bin/bash
# round 5 batch 25: dynamic deal across XCDs in the product library (long D = 64 plain / causal forward launches through fasn_fwd_ws): whole GPU suite, then
# config 5 / a long plain launch / M0 with the in-tree library against the previous one (tools/var/staticdeal = no workspace request)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5a3}; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu_tail.log
for wp in "c5 fwd" "m0 fwd" "c3 fwd"; do
  set -- $wp
  echo "=== $1 $2"; bash tools/ab_libs.sh "bench.py --workload $1 --pass $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes" . staticdeal . staticdeal . staticdeal
done 2>&1 | tee $O/dynamic_deal_product_ab.log
