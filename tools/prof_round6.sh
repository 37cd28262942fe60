#!/bin/bash
# (round 6: prof_round4.sh + the t32 workload)
# usage (via gpurun): tools/prof_round6.sh TAG PART     PART = a (m0 c3 c4 c2 passes) | b (c5 d256, side kernels incl. their PMC pass, bench lines, clocks) | all (a + b without the side kernels: every BASELINE pass, pmc_latest.json and the bench lines from ONE library build)
# Profiles of the product library as shipped (bench.py -> libfasn.so), one directory per workload:pass under gpurun_out/TAG:
# rocprofv3 --kernel-trace --stats, then separate --pmc passes (never together with tracing domains other than kernel dispatch):
# FETCH_SIZE, WRITE_SIZE, two SQ sets (the first one carries GRBM_GUI_ACTIVE: cycles per XCD summed over 8 -> the effective clock of
# the launch). Per-pass fragments gpurun_out/TAG/frag_*.json are merged into pmc_latest.json (keyed by the sha256 of libfasn.so) by part b.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; PART=${2:-a}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {   # run NAME "bench args" LAUNCHES counter-set...   (NAME = workload_pass)
  n=$1; B="python $R/bench.py $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-passes"; L=$3; shift 3; D=$O/$n; mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --stats -d $D/kt -o kt -- $B > $D/kt.log 2>&1
  for set in "$@"; do
    c=$(echo $set | cut -d" " -f1)
    for try in 1 2 3; do   # (round 5: a counter pass now and then stalls right after tool initialisation - three log lines, no kernel ever launched - until its timeout; the config-5 forward FETCH_SIZE pass did so in three calls. Try again instead of leaving a hole in pmc_latest.json)
      rm -rf $D/pmc_$c; timeout 240 rocprofv3 --pmc $set -d $D/pmc_$c -o pmc -- $B --roofline-launches $L --min-timed-ms 0 > $D/pmc_$c.log 2>&1
      [ $(wc -l < $D/pmc_$c.log) -gt 5 ] && break; echo "$n $c: pass stalled (try $try)"
    done
  done
  python3 $R/tools/pmc_summary.py $D fasn_ > $D/summary.txt 2>&1
  python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so $n > /dev/null 2>&1
  find $D -name "*.db" -delete; find $D -type f -size +2M -delete
  echo "$n done: $(grep -c . $D/summary.txt) summary lines"
}
if [ "$PART" = "c5fwd" ]; then   # one workload:pass again (fragment frag_c5_fwd.json)
  run c5_fwd "--workload c5 --pass fwd" 8 FETCH_SIZE WRITE_SIZE "$SQ1"
fi
if [ "$PART" = "c5" ]; then   # only the passes of config 5 (fragments frag_c5_*.json; merge with the others by tools/pmc_to_json.py OUTDIR lib)
  for p in fwd bwd; do run c5_$p "--workload c5 --pass $p" 8 FETCH_SIZE WRITE_SIZE "$SQ1"; done
fi
if [ "$PART" = "a" ] || [ "$PART" = "all" ]; then
  for w in m0 c3 c4; do for p in fwd bwd; do run ${w}_$p "--workload $w --pass $p" 40 FETCH_SIZE WRITE_SIZE "$SQ1" "$SQ2"; done; done
  for p in fwd bwd; do run c2_$p "--workload c2 --pass $p" 40 FETCH_SIZE WRITE_SIZE "$SQ1"; done
  for p in fwd bwd; do run t32_$p "--workload t32 --pass $p" 40 FETCH_SIZE WRITE_SIZE "$SQ1"; done   # (round 6: the shape of the reference's Triton test grid, head dim 32)
fi
if [ "$PART" = "b" ] || [ "$PART" = "all" ]; then
  for p in fwd bwd; do run c5_$p "--workload c5 --pass $p" 8 FETCH_SIZE WRITE_SIZE "$SQ1"; done
  for p in fwd bwd; do run d256_$p "--workload d256 --pass $p" 20 FETCH_SIZE WRITE_SIZE "$SQ1"; done
  python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so > $O/pmc_latest.json
  cp $O/pmc_latest.json $R/profiles/pmc_latest.json
  if [ "$PART" = "b" ]; then
  # side kernels (softmax_n rows, moments, split-K decode, reduced bias gradient) and dropout: kernel trace + one SQ pass
  for s in bench_aux bench_dropout; do
    mkdir -p $O/$s; timeout 600 rocprofv3 --kernel-trace --stats -d $O/$s/kt -o kt -- python $R/tools/$s.py > $O/$s/out.log 2>&1
    timeout 900 rocprofv3 --pmc $SQ1 -d $O/$s/pmc_SQ -o pmc -- python $R/tools/$s.py > $O/$s/pmc.log 2>&1
    python3 $R/tools/pmc_summary.py $O/$s > $O/$s/summary.txt 2>&1; find $O/$s -name "*.db" -delete; find $O/$s -type f -size +2M -delete
  done
  fi
  cd $R
  python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  : > $O/bench_all.jsonl
  for w in m0 c2 c3 c5 c4 d256 t32; do for p in fwd bwd fwdbwd; do
    python bench.py --workload $w --pass $p --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes >> $O/bench_all.jsonl 2>> $O/bench_all.err
  done; done
  python bench.py --workload c1 --steps 50 --warmup 5 >> $O/bench_all.jsonl 2>> $O/bench_all.err
  python - <<PY
import json
for l in open("$O/bench_all.jsonl"):
    d=json.loads(l); r=d["roofline"]
    print("%-64s %8.3f ms/step kernels %8.3f ms  alg %7.1f TF (%.3f) exec %.3f traffic %s" % (d["config"]["workload"][:64], d["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r["frac_executed"], r["traffic"]))
PY
  {
  for a in "8 16 4096 4096 64 1 0 0 20000 0 1.0 0 0" "8 16 4096 4096 64 1 0 0 5000 1 1.0 0 0" "8 16 4096 4096 64 1 1 0 20000 0 1.0 0 0" "4 32 8192 8192 128 1 0 0 2000 0 0.5 4 1" "4 32 8192 8192 128 1 0 0 500 1 0.5 4 1"; do
    echo "== harness bench $a"; bash $R/tools/clock_probe.sh $a
  done
  } > $O/clocks.log 2>&1
fi
du -sh $O
