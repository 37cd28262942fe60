#!/bin/bash
# usage (via gpurun): tools/prof_round3.sh TAG
# Profiles of the product library as shipped (bench.py -> libfasn.so), one directory per workload:pass under gpurun_out/TAG:
# rocprofv3 --kernel-trace --stats, then separate --pmc passes (never together with tracing domains other than kernel dispatch):
# FETCH_SIZE, WRITE_SIZE, two SQ sets. Summaries -> gpurun_out/TAG/pmc_latest.json (keyed by the sha256 of libfasn.so), which is
# copied into profiles/ on the box BEFORE the all-workloads bench lines are taken, so their roofline.traffic is populated.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {   # run NAME "bench args" counter-set...   (NAME = workload_pass)
  n=$1; B="python $R/bench.py $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-passes"; shift 2; D=$O/$n; mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --stats -d $D/kt -o kt -- $B > $D/kt.log 2>&1
  for set in "$@"; do
    c=$(echo $set | cut -d" " -f1); timeout 300 rocprofv3 --pmc $set -d $D/pmc_$c -o pmc -- $B --roofline-launches 40 > $D/pmc_$c.log 2>&1   # (counters are per-dispatch averages; 215 serialised C5 launches ran into the limit on slow boxes)
  done
  python3 $R/tools/pmc_summary.py $D fasn_ > $D/summary.txt 2>&1
  python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so $n > /dev/null 2>&1
  find $D -name "*.db" -delete; find $D -type f -size +2M -delete
}
for w in m0 c3 c4; do for p in fwd bwd; do run ${w}_$p "--workload $w --pass $p" FETCH_SIZE WRITE_SIZE "$SQ1" "$SQ2"; done; done
for w in c2 c5 d256; do for p in fwd bwd; do run ${w}_$p "--workload $w --pass $p" FETCH_SIZE WRITE_SIZE "$SQ1"; done; done
run m0onepass_bwd "--workload m0 --pass bwd --backward-plan one_pass" FETCH_SIZE WRITE_SIZE "$SQ1" "$SQ2"
python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so > $O/pmc_latest.json
cp $O/pmc_latest.json $R/profiles/pmc_latest.json
# kernel traces of the side kernels and the dropout kernels
for s in bench_aux bench_dropout; do
  mkdir -p $O/$s; timeout 600 rocprofv3 --kernel-trace --stats -d $O/$s/kt -o kt -- python $R/tools/$s.py > $O/$s/out.log 2>&1
  python3 $R/tools/pmc_summary.py $O/$s > $O/$s/summary.txt 2>&1; find $O/$s -name "*.db" -delete; find $O/$s -type f -size +2M -delete
done
cd $R
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
    print("%-64s %8.3f ms/step kernels %8.3f ms  alg %7.1f TF (%.3f) exec %.3f traffic %s" % (d["config"]["workload"][:64], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r["frac_executed"], r["traffic"]))
PY
# clock / socket power under load (rocm-smi): the chip runs these kernels at its power cap
{
for a in "8 16 4096 4096 64 1 0 0 20000 0 1.0 0 0" "8 16 4096 4096 64 1 0 0 5000 1 1.0 0 0" "8 16 4096 4096 64 1 1 0 20000 0 1.0 0 0" "4 32 8192 8192 128 1 0 0 2000 0 0.5 0 0" "4 32 8192 8192 128 1 0 0 2000 0 0.5 4 1" "4 32 8192 8192 128 1 0 0 500 1 0.5 4 1"; do
  echo "== harness bench $a"; bash $R/tools/clock_probe.sh $a
done
} > $O/clocks.log 2>&1
du -sh $O
