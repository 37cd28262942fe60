#!/bin/bash
# usage (via gpurun): tools/prof_round.sh TAG   -> gpurun_out/TAG/{fwd_m0,bwd_m0}/summary.txt + bench.py kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1
bash $R/tools/prof_fwd.sh $T/fwd_m0 "8 16 4096 4096 64 1 0 0 50" > /dev/null 2>&1
bash $R/tools/prof_fwd.sh $T/bwd_m0 "8 16 4096 4096 64 1 0 0 20 1" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/bench -o bench -- python $R/bench.py > $R/gpurun_out/$T/bench.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/$T/bench fasn_ > $R/gpurun_out/$T/bench_summary.txt 2>&1
tail -1 $R/gpurun_out/$T/bench.log | cut -c1-400
cat $R/gpurun_out/$T/bench_summary.txt
grep -E "kernel|MFMA_BUSY|BUSY_CYCLES|FETCH_SIZE|WRITE_SIZE|BANK_CONFLICT|GRBM" $R/gpurun_out/$T/fwd_m0/summary.txt | head -30
grep -E "kernel|MFMA_BUSY|SQ_BUSY_CYCLES|BANK_CONFLICT" $R/gpurun_out/$T/bwd_m0/summary.txt | head -30
