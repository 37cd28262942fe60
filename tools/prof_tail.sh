#!/bin/bash
# usage (via gpurun): tools/prof_tail.sh TAG - the tail of tools/prof_round3.sh alone: side kernels, dropout (kernel traces) and the clock / power probe
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in bench_aux bench_dropout; do
  mkdir -p $O/$s; timeout 300 rocprofv3 --kernel-trace --stats -d $O/$s/kt -o kt -- python $R/tools/$s.py > $O/$s/out.log 2>&1
  python3 $R/tools/pmc_summary.py $O/$s > $O/$s/summary.txt 2>&1; find $O/$s -name "*.db" -delete; find $O/$s -type f -size +2M -delete
done
{
for a in "8 16 4096 4096 64 1 0 0 20000 0 1.0 0 0" "8 16 4096 4096 64 1 0 0 5000 1 1.0 0 0" "8 16 4096 4096 64 1 1 0 20000 0 1.0 0 0" "4 32 8192 8192 128 1 0 0 2000 0 0.5 0 0" "4 32 8192 8192 128 1 0 0 2000 0 0.5 4 1" "4 32 8192 8192 128 1 0 0 500 1 0.5 4 1"; do
  echo "== harness bench $a"; timeout 120 bash $R/tools/clock_probe.sh $a
done
} > $O/clocks.log 2>&1
grep -v -E "^[WEI]2026|amdgpu.ids" $O/bench_aux/out.log | head -16; grep -v -E "^[WEI]2026|amdgpu.ids" $O/bench_dropout/out.log | head -6; grep -c sclk $O/clocks.log
