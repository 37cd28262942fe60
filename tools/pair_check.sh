#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-pair}; O=$R/gpurun_out/$T; mkdir -p $O
H=$R/tools/fasn_harness
export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
{
echo "== parity, pairing forced"; FASN_PAIR=1 timeout 900 $H test 0 1 | grep -v "^\[ ok" | tail -5
echo "== parity, shipped rule"; timeout 900 $H test 0 1 | grep -v "^\[ ok" | tail -3
for pm in 0 1 0 1; do
  echo "== C5 pair=$pm"; FASN_PAIR=$pm timeout 120 $H bench 64 16 4096 4096 64 1 1 0 20 1
  echo "== C3 pair=$pm"; FASN_PAIR=$pm timeout 120 $H bench 8 16 4096 4096 64 0 1 0 50 1
  echo "== causal (4,32,8192,128) pair=$pm"; FASN_PAIR=$pm timeout 120 $H bench 4 32 8192 8192 128 1 1 0 10 1
  echo "== causal (8,16,1024,64) pair=$pm"; FASN_PAIR=$pm timeout 120 $H bench 8 16 1024 1024 64 1 1 0 100 1
done
} > $O/log.txt 2>&1
cat $O/log.txt
