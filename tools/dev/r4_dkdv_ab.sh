#!/bin/bash
# round 4: correctness of the software-pipelined dK/dV kernel through the developer harness, then same-box A/B against the round-3 kernel
# (bwd_variant 64 = round-3 dK/dV) and a kernel trace of both.  usage (via gpurun): tools/r4_dkdv_ab.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r4a}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 $H test 0 quick > $O/harness_test.log 2>&1; echo "harness test rc=$?" | tee -a $O/harness_test.log
grep -v '^ok\|^  ok' $O/harness_test.log | tail -30
for rep in 1 2; do for bv in 0 64; do
  echo "== bwd_variant $bv (rep $rep)"
  $H bench 8 16 4096 4096 64 1 0 0 40 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 4096 4096 64 0 1 0 40 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 1024 1024 64 1 0 0 100 1 1.0 0 0 $bv | tail -1
  $H bench 64 16 4096 4096 64 1 1 0 10 1 1.0 0 0 $bv | tail -1
done; done 2>&1 | tee $O/ab.log
for bv in 0 64; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$bv -o kt -- $H bench 8 16 4096 4096 64 1 0 0 40 1 1.0 0 0 $bv > $O/kt_$bv.log 2>&1
  f=$(find $O/kt_$bv -name '*kernel_stats.csv' | head -1); echo "-- kernel stats bwd_variant $bv"; head -8 $f | cut -c1-220
  find $O/kt_$bv -name "*.db" -delete; find $O/kt_$bv -type f -size +2M -delete
done 2>&1 | tee $O/kt.log
