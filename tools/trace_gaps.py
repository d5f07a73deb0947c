#!/usr/bin/env python
"""Idle gaps and low-occupancy stretches of ONE steady-state forward in a rocprofv3 kernel trace: the period between two
consecutive dispatches of a marker kernel (launched once per forward); the union of all kernel intervals is cut into busy /
idle, every idle gap >= min_gap_us is listed with the kernels that end before and start after it, and the time with exactly
one kernel running is summed per kernel family (where the forward is a dependent chain).
usage: trace_gaps.py <kernel_trace.csv> <marker substring> [min_gap_us=15]"""
import collections
import csv
import re
import sys

src, marker = sys.argv[1], sys.argv[2]
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:48]   # noqa: E731
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
marks = [s for s, e, k in rows if marker in k]
gaps = sorted((marks[i + 1] - marks[i], i) for i in range(len(marks) - 1))
_, i = gaps[len(gaps) // 2]
t0, t1 = marks[i], marks[i + 1]
ev = [(s, e, k) for s, e, k in rows if e > t0 and s < t1]
print("forward period %.2f ms, %d dispatches" % ((t1 - t0) / 1e6, len(ev)))
# sweep
pts = []
for s, e, k in ev:
    pts.append((max(s, t0), 1, k))
    pts.append((min(e, t1), -1, k))
pts.sort(key=lambda p: (p[0], p[1]))
active = collections.Counter()
n = 0
last = t0
idle = 0
solo = collections.Counter()
hist = collections.Counter()
last_end_kernel = "-"
for t, d, k in pts:
    dt = t - last
    if dt > 0:
        hist[min(n, 6)] += dt
        if n == 0:
            idle += dt
            if dt >= min_gap * 1e3:
                print("  idle %7.1f us at %8.3f ms   after [%s]  before [%s]" % (dt / 1e3, (last - t0) / 1e6, last_end_kernel, k))
        elif n == 1:
            solo[re.sub(r"<.*", "", next(iter(+active)))] += dt
    if d > 0:
        active[k] += 1
        n += 1
    else:
        active[k] -= 1
        n -= 1
        last_end_kernel = k
    last = t
print("idle total %.1f us" % (idle / 1e3))
print("time by number of kernels in flight: " + "  ".join("%d%s: %.2f ms" % (c, "+" if c == 6 else "", hist[c] / 1e6) for c in sorted(hist)))
print("time with exactly ONE kernel in flight, by family:")
for k, v in solo.most_common(16):
    print("   %8.1f us  %s" % (v / 1e3, k))
