#!/usr/bin/env python
"""Loop basic blocks of one kernel in a `hipcc -S --cuda-device-only` listing, with instruction classes and the VALU opcode
histogram of each: tools/isa_blocks.py file.s kernel-name-substring [min block size]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4fasn") and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks = []; cur = ["entry", []]; blocks.append(cur)
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1) + ("  LOOP" if "Loop" in l else ""), []]; blocks.append(cur)
    elif l.strip() and not l.strip().startswith((";", ".")):
        cur[1].append(l.strip())
for name, ins in blocks:
    c = collections.Counter()
    for x in ins:
        op = x.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith(("s_cbranch", "s_branch")): c["br:" + x.split()[-1]] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c["vmem"] += 1
    if "LOOP" in name and len(ins) >= minsz:
        print(name, len(ins), dict(c))
        h = collections.Counter(x.split()[0] for x in ins if x.startswith("v_") and not x.startswith("v_mfma"))
        print("     ", sorted(h.items(), key=lambda x: -x[1])[:12])
