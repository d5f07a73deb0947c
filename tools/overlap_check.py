#!/usr/bin/env python
"""How much of each kernel family's time inside one steady-state forward overlaps with ANY other kernel (rocprofv3 kernel
trace): tells whether the parallel branches of the captured graph really run side by side.
usage: overlap_check.py <kernel_trace.csv> <marker kernel substring>"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
marks = [s for s, e, k, q in rows if sys.argv[2] in k]
gaps = sorted((marks[i + 1] - marks[i], i) for i in range(len(marks) - 1))
_, i = gaps[len(gaps) // 2]
t0, t1 = marks[i], marks[i + 1]
sel = [(s, e, re.sub(r"<.*", "", re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))), q)
       for s, e, k, q in rows if s >= t0 and s < t1]
fam = collections.defaultdict(lambda: [0.0, 0.0, 0, set()])
for a, (s, e, k, q) in enumerate(sel):
    ov = 0
    # union of the overlaps with the other kernels (sel is sorted by start)
    segs = []
    for b, (s2, e2, k2, q2) in enumerate(sel):
        if b == a or e2 <= s or s2 >= e:
            continue
        segs.append((max(s, s2), min(e, e2)))
    segs.sort()
    cur = s
    for x, y in segs:
        if y > cur:
            ov += y - max(x, cur)
            cur = y
    f = fam[k]
    f[0] += (e - s) / 1e3
    f[1] += ov / 1e3
    f[2] += 1
    f[3].add(q)
print("# one forward period: %.2f ms, %d dispatches" % ((t1 - t0) / 1e6, len(sel)))
print("%-36s %6s %10s %10s %6s  queues" % ("kernel", "calls", "total us", "overlap us", "frac"))
for k, (tot, ov, n, qs) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:30]:
    print("%-36s %6d %10.0f %10.0f %6.2f  %s" % (k[:36], n, tot, ov, ov / tot if tot else 0, ",".join(sorted(qs))))
