"""Fold rocprofv3 counter_collection csvs: per kernel name (substring filter) the mean of each counter per dispatch.
usage: pmc_fold.py <filter> file1.csv [file2.csv ...]"""
import collections
import csv
import sys

flt = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sys.argv[2:]:
    with open(fn) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if flt in k:
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
