#!/bin/bash
# A/B of the D = 128 backward: two-wave dK/dV kernel (default) vs the one-wave kernel (bwd_variant 1)
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness
for bv in 0 1; do
  echo "== bwd_variant $bv"
  $H bench 4 32 8192 8192 128 1 0 0 10 1 1.0 0 0 $bv | tail -1
  $H bench 4 32 8192 8192 128 1 1 0 10 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 2048 2048 128 0 0 0 20 1 1.0 0 0 $bv | tail -1
done
