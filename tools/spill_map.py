#!/usr/bin/env python3
"""Where does a kernel touch scratch? usage: spill_map.py file.s kernel-name-substring
Prints, per basic block: line count, MFMA count, scratch stores / loads, and whether the block is inside a loop."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = end = None
for i, l in enumerate(lines):
    if start is None and l.endswith(':') and pat in l and not l.startswith('.'):
        start = i
    elif start is not None and l.startswith('.Lfunc_end'):
        end = i
        break
body = lines[start:end]
labels = [i for i, l in enumerate(body) if re.match(r'\.LBB\d+_\d+:', l)] + [len(body)]
tot_st = tot_ld = 0
for a, b in zip([0] + labels[:-1], labels):
    blk = body[a:b]
    st = sum('scratch_store' in l for l in blk)
    ld = sum('scratch_load' in l for l in blk)
    tot_st += st; tot_ld += ld
    if st or ld:
        print('%-60s %4d lines mfma %3d  scratch st %3d ld %3d' % (body[a][:60], b - a, sum('v_mfma' in l for l in blk), st, ld))
print('total scratch stores %d loads %d' % (tot_st, tot_ld))
