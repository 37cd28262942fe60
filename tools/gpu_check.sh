#!/bin/bash
# usage (via gpurun): tools/gpu_check.sh TAG [pytest-args]  -> gpurun_out/TAG/{pytest.log,bench_default.json,bench_all.jsonl}
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-1800 $O/bench_default.json
: > $O/bench_all.jsonl
for w in m0 c2 c3 c5 c4; do for p in fwd bwd fwdbwd; do
  python bench.py --workload $w --pass $p --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes >> $O/bench_all.jsonl 2>> $O/bench_all.err
done; done
python - <<PY
import json
for l in open("$O/bench_all.jsonl"):
    d=json.loads(l); r=d["roofline"]
    print("%-70s %8.3f ms/step  kernels %8.3f ms  alg %6.1f TF (%.3f)  exec frac %.3f" % (d["config"]["workload"][:70], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r["frac_executed"]))
PY
