#!/usr/bin/env python3
"""Per-XCD view of a raw forward timeline (FASN_TIMELINE_DUMP of tools/fasn_harness timeline): how many workgroups each XCD ran, when its last one
ended, mean workgroup lifetime - does one slow XCD set the span of a launch whose workgroups are dealt to the XCDs round-robin? usage: timeline_xcd.py dump.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
t0 = a[:, 0].min()
st = (a[:, 0] - t0) * 0.01
en = (a[:, 3] - t0) * 0.01
xcc = a[:, 5].astype(np.int64) & 15
life = en - st
print("workgroups %d, span %.1f us, mean lifetime %.2f us" % (len(a), en.max(), life.mean()))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("xcc %d: %5d workgroups, last end %7.1f us (%.1f %% of the span), mean lifetime %.2f us (%+.1f %% vs all), p10 %.2f p90 %.2f" %
          (x, int(m.sum()), en[m].max(), 100 * en[m].max() / en.max(), life[m].mean(), 100 * (life[m].mean() / life.mean() - 1), *np.percentile(life[m], [10, 90])))
ends = np.array([en[xcc == x].max() for x in sorted(set(xcc.tolist()))])
print("XCD end times: min %.1f max %.1f us: a perfectly balanced deal would end near %.1f us (%.1f %% earlier)" % (ends.min(), ends.max(), ends.mean(), 100 * (1 - ends.mean() / ends.max())))
