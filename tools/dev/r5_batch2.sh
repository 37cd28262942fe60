#!/bin/bash
# round 5 batch 2: (1) config-4 forward, start values built beside the PV MFMAs (in-tree) against at the top of the tile (tools/var/nopipe),
# (2) parity of the bias paths on the in-tree library, (3) causal forward tuning points + timelines (developer harness)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5b}; mkdir -p $O
cd $R
bash tools/ab_libs.sh "bench.py --workload c4 --pass fwd --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes" . nopipe . nopipe 2>&1 | tee $O/c4_pipeseed_ab.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bias or length_paired or config4 or golden_g4 or kernel_path" 2>&1 | tail -5 | tee $O/pytest_bias.log
bash tools/r5_causal.sh ${1:-r5b}
python bench.py --workload c4 --pass fwd --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes > $O/bench_c4_fwd.json 2>&1
