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
    """bench.py's stand-alone roofline launches of one (kernel, grid): the run of `last` consecutive dispatches of that kernel
    with the shortest span (back-to-back launches; inside the captured graph the same kernel overlaps with launches of other
    streams, which stretches its in-graph duration, and the training probe launches it with other epilogues)."""
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            if kernel_substr in r["Kernel_Name"] and g == grid:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    if len(rows) < last:
        return
    best = min(range(len(rows) - last + 1), key=lambda i: rows[i + last - 1][1] - rows[i][0])
    run = rows[best:best + last]
    d = [(e - s_) / 1e3 for s_, e in run]
    gaps = [(run[i + 1][0] - run[i][1]) / 1e3 for i in range(last - 1)]
    with open(dst, "a") as o:
        o.write("#\n# stand-alone launches of %s (grid %d threads): the tightest run of %d consecutive dispatches = bench.py's\n"
                "# roofline measurement (1 warm-up + 20 timed, back to back on one stream)\n" % (kernel_substr, grid, last))
        o.write("# durations_us: %s\n# average duration of the 20 timed: %.1f us; average gap to the next dispatch (end-of-kernel\n"
                "# cache write-back + dispatch): %.1f us; duration + gap = %.1f us = what HIP events around back-to-back launches see\n"
                % (" ".join("%.0f" % x for x in d), sum(d[1:]) / (last - 1), sum(gaps) / len(gaps),
                   sum(d[1:]) / (last - 1) + sum(gaps) / len(gaps)))


if __name__ == "__main__":
    main()
    if len(sys.argv) >= 6:
        isolated(sys.argv[1], sys.argv[2], sys.argv[4], int(sys.argv[5]), 21)
