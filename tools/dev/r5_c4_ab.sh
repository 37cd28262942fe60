#!/bin/bash
# round 5, same-box A/B of the config-4 launches (developer library): rotated second walk of a length pair (FASN_KPROT=0/1), against the
# unpaired schedule (FASN_PAIR=0); HBM fetch and LDS bank-conflict counters per variant.   usage: tools/r5_c4_ab.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; O=$R/gpurun_out/${1:-r5c4}; mkdir -p $O
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
timeout 600 $H test 0 1 > $O/harness_test.log 2>&1; echo "harness test rc=$?"; grep -h 'FAIL\|PASSED\|FAILED' $O/harness_test.log | head -20
C4="4 32 8192 8192 128 1 0 0"
{
for rep in 1 2 3; do
  for kr in 0 1; do echo "== FASN_KPROT=$kr (rep $rep)"; FASN_KPROT=$kr $H bench $C4 60 1 0.5 4 1 | tail -2; done
done
echo "== FASN_PAIR=0 (unpaired schedule)"; FASN_PAIR=0 $H bench $C4 60 1 0.5 4 1 | tail -2
echo "== lengths 8/7/6/5 style (mask_kind 1)"; for kr in 0 1; do FASN_KPROT=$kr $H bench $C4 40 1 0.5 1 1 | tail -2; done
} 2>&1 | tee $O/ab.log
{
for kr in 0 1; do
  echo "== PMC FASN_KPROT=$kr"
  for c in FETCH_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
    FASN_KPROT=$kr bash $R/tools/pmc_one.sh "$c" $C4 6 1 0.5 4 1 | grep -v delta
  done
done
echo "== PMC FASN_PAIR=0"; FASN_PAIR=0 bash $R/tools/pmc_one.sh FETCH_SIZE $C4 6 1 0.5 4 1 | grep -v delta
} 2>&1 | tee $O/pmc.log
