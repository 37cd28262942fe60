#!/bin/bash
# register / spill summary of every kernel of one .hip file: tools/kres.sh file.hip [extra hipcc flags]
f=$1; shift
cd $(dirname $0)/../flash-attention-softmax-n_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-value -Wno-inline-asm "$@" -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
name=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); d[name]={}
    for k in ('VGPRs','AGPRs','SGPRs Spill','VGPRs Spill','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]'):
        m=re.search(r'remark: \s*'+k+r': (\d+)',l)
        if m and name: d[name][k]=int(m.group(1))
for n,v in d.items():
    print('%-110s vgpr %3d agpr %3d occ %d sspill %3d vspill %3d' % (n[:110], v.get('VGPRs',0), v.get('AGPRs',0), v.get('Occupancy \[waves/SIMD\]',0), v.get('SGPRs Spill',0), v.get('VGPRs Spill',0)))
"
