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


def timeline(src, dst, marker, bin_us=1000.0):
    """One steady-state forward (the interval between two consecutive dispatches of `marker`, a kernel launched once per
    forward), cut into bins: per bin the kernel families by busy time summed over the streams."""
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    marks = [s_ for s_, e, k in rows if marker in k]
    if len(marks) < 4:
        return
    gaps = sorted((marks[i + 1] - marks[i], i) for i in range(len(marks) - 1))
    ref = gaps[len(gaps) // 2 + len(gaps) // 4][0]                  # median of the longer half (a marker launched twice in a row
    gaps = [g for g in gaps if 0.5 * ref <= g[0] <= 1.5 * ref]     #  leaves short gaps, a pause between bench phases a long one)
    _, i = gaps[len(gaps) // 2]                                   # a median-length period
    t0, t1 = marks[i], marks[i + 1]
    nb = int((t1 - t0) / 1e3 / bin_us) + 1
    bins = [collections.defaultdict(float) for _ in range(nb)]
    for s_, e, k in rows:
        if e <= t0 or s_ >= t1:
            continue
        a, b = max(s_, t0), min(e, t1)
        fam = re.sub(r"<.*", "", k)
        j = int((a - t0) / 1e3 / bin_us)
        while a < b and j < nb:
            edge = t0 + int((j + 1) * bin_us * 1e3)
            bins[j][fam] += (min(b, edge) - a) / 1e3
            a, j = edge, j + 1
    with open(dst, "w") as o:
        o.write("# one forward period (%.2f ms) between two dispatches of %s; per %.0f us bin: busy us per kernel family\n"
                % ((t1 - t0) / 1e6, marker, bin_us))
        for j, b in enumerate(bins):
            top = sorted(b.items(), key=lambda kv: -kv[1])[:5]
            o.write("%5.1f ms  busy %6.0f us  %s\n" % (j * bin_us / 1e3, sum(b.values()),
                                                       "  ".join("%s %.0f" % (k, v) for k, v in top)))


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


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--timeline":
    timeline(sys.argv[2], sys.argv[3], sys.argv[4])
    sys.exit(0)
if __name__ == "__main__":
    main()
    if len(sys.argv) >= 6:
        isolated(sys.argv[1], sys.argv[2], sys.argv[4], int(sys.argv[5]), 21)
