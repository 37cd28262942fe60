#!/usr/bin/env python
"""Instruction histogram of the hottest loop(s) of one kernel in a gfx950 .s file.
usage: asm_loop.py file.s kernel_substring"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm"))
body = src[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print(f"kernel lines {len(body)}; backward branches (loops): {[(a, b, b - a) for a, b in loops]}")
def cat(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rcp"): return "TRANS"
    if op.startswith("v_"): return "VALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "VMEM"
    if op.startswith("s_waitcnt"): return "WAIT"
    if op.startswith("s_barrier"): return "BARRIER"
    if op.startswith("s_"): return "SALU"
    return "OTHER"
for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    ops = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(cat(o) for o in ops)
    print(f"loop lines {a}..{b}: {dict(c)}")
    d = collections.Counter(ops)
    print("  " + ", ".join(f"{k}:{v}" for k, v in d.most_common(28)))
