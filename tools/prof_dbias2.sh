#!/bin/bash
# instruction-fetch / scalar-cache / branch counters of the bias-gradient kernel (tools/bench_dbias.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pd2; mkdir -p /tmp/pd2
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d /tmp/pd2/p$i -o pmc -- python $R/tools/bench_dbias.py > /tmp/pd2/log$i.txt 2>&1 || echo "set $i failed: $set"
done
python3 $R/tools/pmc_summary.py /tmp/pd2 dbias | sed 's/.*\] //' | cut -c1-150
