#!/bin/bash
# round 5 batch 4: the in-tree library after the causal change - parity on the causal / golden / property tests, bench lines of the BASELINE workloads
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5d}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "causal or golden or analytic or properties or fuzz or random" 2>&1 | tail -4 | tee $O/pytest_causal.log
: > $O/bench.jsonl
for w in m0 c3 c5 c2; do python bench.py --workload $w --pass fwd --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes >> $O/bench.jsonl 2>> $O/bench.err; done
python - <<PY
import json
for l in open("$O/bench.jsonl"):
    d=json.loads(l); r=d["roofline"]
    print("%-64s %8.3f ms/step kernels %8.3f ms  alg %7.1f TF (%.3f)  %s" % (d["config"]["workload"][:64], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r["kernels"]))
PY
