#!/usr/bin/env python
"""Instruction mix of one kernel in a hipcc -S --cuda-device-only listing: tools/isa_mix.py file.s substring [substring...]
(counts static instructions by class; per-role loops of the wave-specialised kernels show up as separate basic-block ranges)."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
for key in sys.argv[2:]:
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4fasn") and key in l and l.rstrip().split(":")[0].endswith("E"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    c = collections.Counter()
    for l in lines[start + 1:end]:
        l = l.strip()
        m = re.match(r"([a-z_0-9]+)\s", l + " ")
        if not m or l.startswith((".", ";")) or l.endswith(":"):
            continue
        op = m.group(1)
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("ds_"): c["lds"] += 1; c["d:" + op] += 1
        elif op.startswith("v_"): c["valu"] += 1; c["v:" + op] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith(("buffer", "global", "scratch")): c["vmem"] += 1; c["m:" + op] += 1
    print(key, {k: v for k, v in c.items() if ":" not in k})
    print("  valu:", sorted([(v, k[2:]) for k, v in c.items() if k.startswith("v:")], reverse=True)[:22])
    print("  lds :", sorted([(v, k[2:]) for k, v in c.items() if k.startswith("d:")], reverse=True))
    print("  vmem:", sorted([(v, k[2:]) for k, v in c.items() if k.startswith("m:")], reverse=True))
