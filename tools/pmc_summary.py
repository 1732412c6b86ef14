#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv rows per kernel (name prefix filter)."""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if pat in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k)
        for c, x in v.items():
            print(f"    {c:42s} per launch {x / n[(k, c)]:16.1f}   launches {n[(k, c)]}")
