#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-c4}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
{
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
timeout 900 tools/fasn_harness test 0 1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bias or alibi or mask or golden" 2>&1 | tail -5
timeout 600 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline
timeout 600 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes
} > $O/log.txt 2>&1
cat $O/log.txt
