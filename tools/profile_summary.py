#!/usr/bin/env python
"""Fold a rocprofv3 --kernel-trace CSV into the per-(kernel, grid) table committed under profiles/.
usage: profile_summary.py <kernel_trace.csv> <out.txt> "<command that was profiled>" """
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    agg = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    with open(src) as f:
        for r in csv.DictReader(f):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
                   int(r["Workgroup_Size_X"]))
            agg[key][0] += 1
            agg[key][1] += d
            total += d
    with open(dst, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats of `%s`, grouped by (kernel, grid threads, block)\n" % cmd)
        o.write("# total GPU kernel time %.1f ms over %d dispatches\n" % (total / 1e3, sum(v[0] for v in agg.values())))
        o.write("# total_ms  share   calls  avg_us  grid  block  kernel\n")
        for (k, g, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write("%9.3f %6.2f%% %6d %8.1f %8d %5d  %s\n" % (t / 1e3, 100 * t / total, n, t / n, g, b, k))


if __name__ == "__main__":
    main()
