#!/bin/bash
# usage (via gpurun): tools/prof_round.sh TAG
# For the product library as shipped (bench.py -> libfasn.so): kernel trace + separate FETCH_SIZE / WRITE_SIZE / MFMA passes of the
# headline workloads, summarised into gpurun_out/TAG/pmc_latest.json keyed by the sha256 of libfasn.so (bench.py reports
# roofline.traffic only when that hash matches the library it runs). Counter passes never share a run with tracing domains
# other than kernel dispatch.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wp in "m0 fwd" "m0 bwd" "c3 fwd" "c4 fwd" "c4 bwd" "c5 fwd"; do
  set -- $wp; w=$1; p=$2; D=$O/${w}_${p}; mkdir -p $D
  B="python $R/bench.py --workload $w --pass $p --steps 5 --warmup 2 --no-cpu-baseline --no-extra-passes"
  rocprofv3 --kernel-trace --stats -d $D/kt -o kt -- $B > $D/kt.log 2>&1
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
    n=$(echo $set | cut -d" " -f1); rocprofv3 --pmc $set -d $D/pmc_$n -o pmc -- $B > $D/pmc_$n.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $D fasn_ > $D/summary.txt 2>&1
done
python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so > $O/pmc_latest.json
cat $O/pmc_latest.json
