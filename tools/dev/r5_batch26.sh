#!/bin/bash
# round 5 batch 26: dynamic deal across XCDs, forward and backward, in the product library against tools/var/staticdeal (no workspace request): GPU suite, then C5 / M0 / C3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5b3}; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu_tail.log
for wp in "c5 bwd" "m0 bwd" "c5 fwd" "c3 bwd"; do
  set -- $wp
  echo "=== $1 $2"; bash tools/ab_libs.sh "bench.py --workload $1 --pass $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes" . staticdeal . staticdeal
done 2>&1 | tee $O/dynamic_deal_product_ab.log
