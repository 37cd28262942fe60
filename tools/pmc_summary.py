#!/usr/bin/env python
"""Summarise rocprofv3 rocpd .db outputs: per-kernel average duration (kernel trace) and per-kernel mean counter values.
usage: pmc_summary.py DIR [kernel_substring]   (DIR is searched recursively for *_results.db)"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else ""
for db in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
    c = sqlite3.connect(db)
    rel = os.path.relpath(db, root)
    try:
        rows = c.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels group by name").fetchall()
        for name, cnt, avg, mn, mx in rows:
            if key in name:
                print(f"[{rel}] kernel {name[:90]}: calls={cnt} avg={avg/1e3:.1f}us min={mn/1e3:.1f}us max={mx/1e3:.1f}us")
    except Exception as e:
        print(rel, "kernels:", e)
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        for kn, cn, val, cnt in rows:
            if key in kn:
                print(f"[{rel}] {kn[:50]:50s} {cn:28s} {val:18.1f}  (n={cnt})")
    except Exception as e:
        pass
