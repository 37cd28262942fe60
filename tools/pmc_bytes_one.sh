#!/bin/bash
# usage (via gpurun): tools/pmc_bytes_one.sh TAG WORKLOAD PASS [launches] - only the FETCH_SIZE / WRITE_SIZE passes of one workload:pass, short roofline loop
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; W=$2; P=$3; L=${4:-10}; O=$R/gpurun_out/$T; D=$O/${W}_$P; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload $W --pass $P --steps 2 --warmup 1 --no-cpu-baseline --no-extra-passes --roofline-launches $L"
for c in FETCH_SIZE WRITE_SIZE; do
  ( time timeout 330 rocprofv3 --pmc $c -d $D/pmc_$c -o pmc -- $B > $D/pmc_$c.log 2>&1 ) 2>&1 | grep real
done
python3 $R/tools/pmc_summary.py $D fasn_ 2>&1 | grep -E "FETCH_SIZE|WRITE_SIZE" | grep -v calls= | cut -c1-160
find $D -name "*.db" -delete; find $D -type f -size +2M -delete
