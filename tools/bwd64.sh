#!/bin/bash
# A/B of the D = 64 backward: bwd_variant 0 = two-wave dQ and dK/dV, 1 = one-wave dK/dV, 2 = one-wave dQ, 3 = both one-wave
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness
for bv in ${1:-0 1 2 3}; do
  echo "== bwd_variant $bv"
  $H bench 8 16 4096 4096 64 1 0 0 50 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 4096 4096 64 0 1 0 50 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 1024 1024 64 1 0 0 100 1 1.0 0 0 $bv | tail -1
done
