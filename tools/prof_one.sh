#!/bin/bash
# usage (via gpurun): tools/prof_one.sh TAG WORKLOAD PASS   - one workload:pass of tools/prof_round3.sh again (kernel trace + FETCH / WRITE / SQ
# passes) -> gpurun_out/TAG/frag_WORKLOAD_PASS.json, to be merged into pmc_latest.json of the same library
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; W=$2; P=$3; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS"
n=${W}_$P; B="python $R/bench.py --workload $W --pass $P --steps 3 --warmup 1 --no-cpu-baseline --no-extra-passes"; D=$O/$n; mkdir -p $D
timeout 600 rocprofv3 --kernel-trace --stats -d $D/kt -o kt -- $B > $D/kt.log 2>&1
for set in FETCH_SIZE WRITE_SIZE "$SQ1"; do
  c=$(echo $set | cut -d" " -f1); timeout 600 rocprofv3 --pmc $set -d $D/pmc_$c -o pmc -- $B --roofline-launches 40 > $D/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_summary.py $D fasn_ > $D/summary.txt 2>&1
python3 $R/tools/pmc_to_json.py $O $R/flash-attention-softmax-n_amd/libfasn.so $n > /dev/null 2>&1
find $D -name "*.db" -delete; find $D -type f -size +2M -delete
cat $O/frag_$n.json
