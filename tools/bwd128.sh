#!/bin/bash
# A/B of the D = 128 backward: bwd_variant 0 = two-wave dQ and dK/dV kernels, 1 = one-wave dK/dV, 2 = one-wave dQ, 3 = both one-wave
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness
for bv in ${1:-0 2}; do
  echo "== bwd_variant $bv"
  $H bench 4 32 8192 8192 128 1 0 0 10 1 1.0 0 0 $bv | tail -1
  $H bench 4 32 8192 8192 128 1 1 0 10 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 2048 2048 128 0 0 0 20 1 1.0 0 0 $bv | tail -1
  $H bench 4 32 8192 8192 128 1 0 0 10 1 0.5 1 1 $bv | tail -1
done
