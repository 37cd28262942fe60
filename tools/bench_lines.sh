#!/bin/bash
# usage (via gpurun): tools/bench_lines.sh TAG  -> gpurun_out/TAG/bench_default.json + bench_all.jsonl (every workload x pass, c1, one-pass backward)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
: > $O/bench_all.jsonl
for w in m0 c2 c3 c5 c4 d256; do for p in fwd bwd fwdbwd; do
  python bench.py --workload $w --pass $p --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes >> $O/bench_all.jsonl 2>> $O/bench_all.err
done; done
python bench.py --workload c1 --steps 50 --warmup 5 >> $O/bench_all.jsonl 2>> $O/bench_all.err
python bench.py --workload m0 --pass bwd --backward-plan one_pass --steps 20 --warmup 5 --no-cpu-baseline >> $O/bench_all.jsonl 2>> $O/bench_all.err
python - <<PY
import json
for l in open("$O/bench_all.jsonl"):
    d=json.loads(l); r=d["roofline"]
    print("%-64s %8.3f ms/step kernels %8.3f ms  alg %7.1f TF (%.3f) exec %.3f traffic %s %s" % (d["config"]["workload"][:64], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r["frac_executed"], r["traffic"], r.get("under_load")))
PY
