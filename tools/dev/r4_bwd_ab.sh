#!/bin/bash
# round 4: software-pipelined D = 64 backward kernels (fasn_bwd_pipe.h) through the developer harness: correctness (default launch rule and
# forced causal pairing), same-box A/B against the round-3 kernels (bwd_variant bit 6 = round-3 dK/dV, bit 7 = round-3 dQ), kernel trace, PMC.
# usage (via gpurun): tools/r4_bwd_ab.sh TAG [notest]
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r4b}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$2" != "notest" ]; then
  timeout 400 $H test 0 1 > $O/harness_test.log 2>&1; echo "harness test rc=$?" | tee -a $O/harness_test.log
  FASN_PAIR=1 timeout 400 $H test 0 1 > $O/harness_test_pair.log 2>&1; echo "harness test (forced pairing) rc=$?" | tee -a $O/harness_test_pair.log
  grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test.log $O/harness_test_pair.log | head -40
fi
for rep in 1 2; do for bv in 0 64 128 192; do
  echo "== bwd_variant $bv (rep $rep)"
  $H bench 8 16 4096 4096 64 1 0 0 40 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 4096 4096 64 0 1 0 40 1 1.0 0 0 $bv | tail -1
  $H bench 8 16 1024 1024 64 1 0 0 100 1 1.0 0 0 $bv | tail -1
  $H bench 64 16 4096 4096 64 1 1 0 10 1 1.0 0 0 $bv | tail -1
done; done 2>&1 | tee $O/ab.log
for bv in 0 192; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$bv -o kt -- $H bench 8 16 4096 4096 64 1 0 0 40 1 1.0 0 0 $bv > $O/kt_$bv.log 2>&1
  echo "-- kernel stats bwd_variant $bv"; python3 $R/tools/pmc_summary.py $O/kt_$bv fasn_ | sed 's/.*\] //' | cut -c1-200
  find $O/kt_$bv -name "*.db" -delete; find $O/kt_$bv -type f -size +2M -delete
done 2>&1 | tee $O/kt.log
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ3="SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VALU_TRANS"
for bv in 0 192; do
  echo "=== PMC bwd_variant $bv"
  for set in "$SQ1" "$SQ2" "$SQ3"; do $R/tools/pmc_one.sh "$set" 8 16 4096 4096 64 1 0 0 3 1 1.0 0 0 $bv | grep -v "fwd_kernel\|delta"; done
done 2>&1 | tee $O/pmc.log
