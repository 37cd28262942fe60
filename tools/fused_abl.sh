#!/bin/bash
# one-pass backward: ablations at M0 (bwd_variant >> 4: 1 no atomics, 2 no dQ GEMM, 3 no dQ GEMM and no dS image), 4 = split kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness
for bv in ${1:-0 16 32 48 4}; do
  echo "== bwd_variant $bv"
  $H bench 8 16 4096 4096 64 1 0 0 30 1 1.0 0 0 $bv | tail -1
done
