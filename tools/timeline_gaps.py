#!/usr/bin/env python3
"""Per-CU view of a raw forward timeline (FASN_TIMELINE_DUMP of tools/fasn_harness timeline): gaps between consecutive workgroups of a CU,
start order versus workgroup id, and the first entries of a few CUs. usage: timeline_gaps.py dump.bin [workgroups_per_cu]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = a[:, 0].min()
st = (a[:, 0] - t0) * 0.01
en = (a[:, 3] - t0) * 0.01
hw = a[:, 4].astype(np.int64)
xcc = a[:, 5].astype(np.int64) & 15
cu = (xcc << 16) | (hw & 0xff00) | (((hw >> 13) & 7) << 4) | ((hw >> 12) & 1)
ids = np.arange(len(a))
print("workgroups", len(a), "CUs", len(set(cu.tolist())), "span %.1f us" % en.max())
print("wgid & 7 == xcc for %.1f%% of the workgroups" % (100.0 * np.mean((ids & 7) == xcc)))
order = np.argsort(st, kind="stable")
print("start order inversions vs workgroup id: %d of %d adjacent pairs" % (int(np.sum(np.diff(order) < 0)), len(a) - 1))
gaps = []
for c in sorted(set(cu.tolist())):
    m = np.where(cu == c)[0]
    m = m[np.argsort(st[m])]
    if occ == 1:
        gaps.extend((st[m][1:] - en[m][:-1]).tolist())
gaps = np.array(gaps)
if len(gaps):
    print("gap between a CU's consecutive workgroups: mean %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f us; negative (overlap) %d" %
          (gaps.mean(), *np.percentile(gaps, [10, 50, 90, 99]), gaps.max(), int((gaps < 0).sum())))
for c in sorted(set(cu.tolist()))[:3]:
    m = np.where(cu == c)[0]
    m = m[np.argsort(st[m])]
    print("CU %x:" % c, " ".join("[%d %.0f-%.0f n%d]" % (i, st[i], en[i], a[i, 6]) for i in m[:10]))
# dispatch lag: start time of workgroup i minus the earliest end among the previous... (simple view: start times vs id per XCD)
for x in range(8):
    m = np.where(xcc == x)[0]
    m = m[np.argsort(m)]
    print("xcc", x, "first starts by id:", " ".join("%.0f" % v for v in st[m][28:44]))
