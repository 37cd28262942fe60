#!/usr/bin/env python3
"""Where do the kernels of a `hipcc -S` listing touch scratch: per kernel the spill count, the scratch loads / stores in total and those
INSIDE a loop (between a backward branch's target label and the branch). A reload inside a tile loop is a vmcnt wait that drains the
K/V prefetch; a spill parked across the loop and reloaded in the epilogue costs nothing measurable.
usage: spill_where.py file.s [substring]"""
import re
import subprocess
import sys


def main():
    src = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    meta = dict(re.findall(r"\.name:\s+(_Z\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", "\n".join(src)))
    names = [src[i][:-1] for i in starts]
    try:
        dm = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    except Exception:
        dm = names
    for i, n, d in zip(starts, names, dm):
        if want not in d and want not in n:
            continue
        end = next(j for j in range(i, len(src)) if src[j].strip().startswith("s_endpgm"))
        body = src[i:end + 1]
        labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for k, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        sc = [(k, "st" if "scratch_store" in l else "ld") for k, l in enumerate(body) if "scratch_store" in l or "scratch_load" in l]
        inl = [(k, t) for k, t in sc if any(a <= k <= b for a, b in loops)]
        big = max(loops, key=lambda t: t[1] - t[0]) if loops else None
        inbig = [(k, t) for k, t in sc if big and big[0] <= k <= big[1]]
        print(f"spill {meta.get(n, '?'):>4}  scratch ops {len(sc):3d}  in loops {len(inl):3d}  in the longest loop {len(inbig):3d} (lines {big})  {d[:120]}")


if __name__ == "__main__":
    main()
