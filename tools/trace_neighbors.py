#!/usr/bin/env python
"""Which kernels run right before / after a given kernel in a rocprofv3 kernel trace (per stream order): finds the host call
site behind anonymous runtime kernels such as __amd_rocclr_copyBuffer.
usage: trace_neighbors.py <kernel_trace.csv> <substring> [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"<.*", "", n.replace("(anonymous namespace)::", "")))[:50]   # noqa: E731
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r.get("Queue_Id", "0")].append(r)
pairs = collections.Counter()
for q, rs in by_q.items():
    for i, r in enumerate(rs):
        if key in r["Kernel_Name"]:
            prev = short(rs[i - 1]["Kernel_Name"]) if i else "-"
            nxt = short(rs[i + 1]["Kernel_Name"]) if i + 1 < len(rs) else "-"
            pairs[(prev, nxt)] += 1
for (a, b), n in pairs.most_common(top):
    print("%6d  %-50s -> [%s] -> %s" % (n, a, key, b))
