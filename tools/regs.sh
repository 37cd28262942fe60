#!/bin/bash
# usage: tools/regs.sh file.hip  -> compact per-kernel register / spill / occupancy table
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/regs_$$.o "$@" 2>&1 \
 | grep -E "error|Function Name|VGPRs:|AGPRs|VGPRs Spill|Occupancy|ScratchSize" \
 | sed -E 's/.*remark: +//; s/\[-Rpass.*//; s/Function Name: /\n/; s/_ZN4fasn[0-9]+//; s/EEEvNS_[0-9A-Za-z]+E//' | tr -s ' \n' ' ' | sed 's/ I/\nI/g; s/ fasn_/\nfasn_/g'
echo; rm -f /tmp/regs_$$.o
