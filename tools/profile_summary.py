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


def isolated(src, dst, kernel_substr, grid, last):
    """The `last` final dispatches of one (kernel, grid): bench.py's stand-alone roofline launches, which follow the timed
    region (inside the captured graph the same kernel overlaps with launches of other streams, which stretches its
    in-graph duration)."""
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            if kernel_substr in r["Kernel_Name"] and g == grid:
                rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    d = [x[1] for x in rows[-last:]]
    with open(dst, "a") as o:
        o.write("#\n# stand-alone launches of %s (grid %d threads): the last %d dispatches of the trace = bench.py's roofline\n"
                "# measurement (20 timed + 1 warm-up); in-graph dispatches of the same kernel overlap with other streams\n"
                % (kernel_substr, grid, last))
        o.write("# durations_us: %s\n# average of the 20 timed: %.1f us\n" % (" ".join("%.0f" % x for x in d), sum(d[-20:]) / 20))


if __name__ == "__main__":
    main()
    if len(sys.argv) >= 6:
        isolated(sys.argv[1], sys.argv[2], sys.argv[4], int(sys.argv[5]), 21)
