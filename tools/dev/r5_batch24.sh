#!/bin/bash
# round 5 batch 24: experiment - dynamic deal of the plain forward's items across XCDs (developer library, env FASN_XQ = surplus workgroups per XCD)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5z2}; mkdir -p $O
cd $R/tools
{
FASN_XQ=8 timeout 300 ./fasn_harness test 0 1 2>&1 | tail -1
for rep in 1 2 3; do
  for shape in "8 16 4096 4096" "32 16 4096 4096" "8 16 8192 8192"; do
    set -- $shape
    echo -n "static deal      : "; timeout 120 ./fasn_harness bench $1 $2 $3 $4 64 1 0 0 1000 2>&1 | tail -1
    for x in 0 8 16 32; do echo -n "dynamic, surplus $x: "; FASN_XQ=$x timeout 120 ./fasn_harness bench $1 $2 $3 $4 64 1 0 0 1000 2>&1 | tail -1; done
  done
done
FASN_TIMELINE_DUMP=/tmp/tl.bin ./fasn_harness timeline 8 16 4096 4096 64 1 0 0 > /dev/null 2>&1; python timeline_xcd.py /tmp/tl.bin
FASN_XQ=16 FASN_TIMELINE_DUMP=/tmp/tl2.bin ./fasn_harness timeline 8 16 4096 4096 64 1 0 0 > /dev/null 2>&1; python timeline_xcd.py /tmp/tl2.bin
} 2>&1 | tee $O/dynamic_xcd_deal.log
