#!/bin/bash
# usage (via gpurun): tools/prof_round.sh TAG [quick]
# For the product library as shipped (bench.py -> libfasn.so): kernel trace + separate FETCH_SIZE / WRITE_SIZE / SQ passes of the
# headline workloads, summarised into gpurun_out/TAG/pmc_latest.json keyed by the sha256 of libfasn.so (bench.py reports
# roofline.traffic only when that hash matches the library it runs). Counter passes never share a run with tracing domains
# other than kernel dispatch. Raw rocpd databases are summarised per pass and deleted (gpurun_out/ is capped at 64 MiB).
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; Q=$2; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {   # run WORKLOAD PASS counter-set...
  w=$1; p=$2; shift 2; D=$O/${w}_${p}; mkdir -p $D
  B="python $R/bench.py --workload $w --pass $p --steps 3 --warmup 1 --no-cpu-baseline --no-extra-passes"
  timeout 300 rocprofv3 --kernel-trace --stats -d $D/kt -o kt -- $B > $D/kt.log 2>&1
  for set in "$@"; do
    n=$(echo $set | cut -d" " -f1); timeout 300 rocprofv3 --pmc $set -d $D/pmc_$n -o pmc -- $B > $D/pmc_$n.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $D fasn_ > $D/summary.txt 2>&1
  python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so ${w}_${p} > /dev/null 2>&1
  find $D -name "*.db" -delete; find $D -type f -size +2M -delete
}
run m0 fwd FETCH_SIZE WRITE_SIZE "$SQ1" "$SQ2"
run c4 fwd FETCH_SIZE WRITE_SIZE "$SQ1"
if [ -z "$Q" ]; then
  run m0 bwd FETCH_SIZE WRITE_SIZE
  run c4 bwd FETCH_SIZE WRITE_SIZE "$SQ2"
  run c3 fwd FETCH_SIZE WRITE_SIZE
fi
python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so > $O/pmc_latest.json
cat $O/pmc_latest.json; du -sh $O
