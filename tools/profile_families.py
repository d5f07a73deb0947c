#!/usr/bin/env python
"""Fold a *_by_grid.txt table (tools/profile_summary.py) into per-kernel-family totals per step.
usage: profile_families.py <by_grid.txt> <steps profiled incl. warm-up>"""
import collections
import re
import sys

steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    f = line.split(None, 6)
    name = f[6].strip()
    m = re.search(r"(nhwc_\w+_kernel|bn_\w+_kernel|\w+_kernel)", name)
    name = m.group(1) if m and not name.startswith("at::") else re.sub(r"<.*", "", name)
    agg[name[:70]][0] += int(f[2])
    agg[name[:70]][1] += float(f[0])
tot = sum(v[1] for v in agg.values())
print("# %.1f ms of kernel time per step" % (tot / steps))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%8.2f ms/step %5.1f%% %6d calls/step  %s" % (t / steps, 100 * t / tot, round(n / steps), k))
