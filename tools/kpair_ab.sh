#!/bin/bash
# same-box A/B of the length-paired batch schedule (developer library): FASN_PAIR=0 plain schedule, 1 paired whatever the lengths, unset = shipped rule
R=${GRAFT_REPO_ROOT:-/root/repo}; H=$R/tools/fasn_harness; export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
BWD=${1:-0}
for mk in 1 4 3; do for pm in 0 1; do
  echo "== mask_kind $mk FASN_PAIR=$pm"
  FASN_PAIR=$pm $H bench 4 32 8192 8192 128 1 0 0 20 $BWD 0.5 $mk 1 | tail -1
  FASN_PAIR=$pm bash $R/tools/pmc_one.sh FETCH_SIZE 4 32 8192 8192 128 1 0 0 5 $BWD 0.5 $mk 1 | grep -v delta
done; done
if [ "$BWD" = 0 ]; then
for mk in 1 3; do for pm in 0 1; do
  echo "== timeline mask_kind $mk FASN_PAIR=$pm"; FASN_PAIR=$pm $H timeline 4 32 8192 8192 128 1 0 0 0.5 $mk 1 | head -4 | cut -c1-250
done; done
fi
