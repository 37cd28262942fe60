#!/usr/bin/env python3
"""Rewrite tests/golden/spill_allowance.json from the library as built: every kernel that still spills, with its count - but never MORE
than the recorded allowance (a kernel that got worse, or a new kernel that spills, is reported and left out: the gate then fails on it).
usage: tools/spill_allowance_update.py [--print]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import spill_map  # noqa: E402

LIB = os.path.join(ROOT, "flash-attention-softmax-n_amd", "libfasn.so")
ALLOW = os.path.join(ROOT, "tests", "golden", "spill_allowance.json")


def main():
    table = spill_map.kernel_table(LIB)
    old = json.load(open(ALLOW))
    new, worse = {}, []
    for n, v in sorted(table.items()):
        s = v.get("spill", 0) if v.get("scratch", 0) > 0 else 0   # (registers parked in accumulation registers - scratch size 0 - are not memory traffic: tests/test_spill_gate.py)
        if not s:
            continue
        if s > old.get(n, 0):
            worse.append((n, s, old.get(n, 0)))
            if n in old:
                new[n] = old[n]
        else:
            new[n] = s
    names = spill_map.demangle(sorted(new))
    for n in sorted(new, key=lambda k: -new[k]):
        print(f"{new[n]:4d} (was {old.get(n, 0):4d})  {names[n][:150]}")
    gone = sorted(set(old) - set(new))
    print(f"{len(new)} kernels with an allowance (was {len(old)}); {len(gone)} entries dropped (no longer spilling or no longer built)")
    for n, s, o in worse:
        print(f"WORSE / NEW: {s} > {o}  {n}")
    if "--print" not in sys.argv:
        with open(ALLOW, "w") as f:
            f.write("{\n" + ",\n".join(f'"{n}": {new[n]}' for n in sorted(new)) + "\n}\n")
    return 1 if worse else 0


if __name__ == "__main__":
    sys.exit(main())
