#!/bin/bash
# correctness battery with the one-pass backward requested + A/B against the split kernels (bwd_variant bit 3 = one pass)
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness
$H test 0 1 1 2>&1 | tail -30
for bv in 8 0; do
  echo "== bwd_variant $bv"
  $H bench 8 16 4096 4096 64 1 0 0 30 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 4096 4096 64 0 1 0 30 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 1024 1024 64 1 0 0 100 1 1.0 0 0 $bv | tail -1
  $H bench 64 16 4096 4096 64 1 1 0 5 1 1.0 0 0 $bv | tail -1
done
