#!/bin/bash
# round 4 quick same-box A/B of backward variants through the developer harness: tools/r4_quick.sh TAG "variants" [test-variant]
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r4q}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$3" ]; then
  FASN_BWDV=$3 timeout 400 $H test 0 1 > $O/harness_test_$3.log 2>&1; echo "harness test (bwd_variant $3) rc=$?"
  grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test_$3.log | head -20
fi
for rep in 1 2; do for bv in $2; do
  echo "== bwd_variant $bv (rep $rep)"
  $H bench 8 16 4096 4096 64 1 0 0 40 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 4096 4096 64 0 1 0 40 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 1024 1024 64 1 0 0 100 1 1.0 0 0 $bv | tail -1
  $H bench 64 16 4096 4096 64 1 1 0 10 1 1.0 0 0 $bv | tail -1
done; done 2>&1 | tee $O/ab.log
for bv in $2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$bv -o kt -- $H bench 8 16 4096 4096 64 1 0 0 40 1 1.0 0 0 $bv > $O/kt_$bv.log 2>&1
  echo "-- kernel stats bwd_variant $bv"; python3 $R/tools/pmc_summary.py $O/kt_$bv fasn_bwd | sed 's/.*\] //' | cut -c1-200
  find $O/kt_$bv -name "*.db" -delete; find $O/kt_$bv -type f -size +2M -delete
done 2>&1 | tee $O/kt.log
