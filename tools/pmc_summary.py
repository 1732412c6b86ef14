#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv rows per (kernel, grid size): launches of one kernel with different grids (the
K = 1536 and K = 512 layers, the tiny parity pass) are different populations and are NOT averaged together (round 1 keyed on
a 48-character name prefix and divided by the wrong launch count).

usage: pmc_summary.py <dir> [name substring]"""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if pat not in name:
            continue
        k = (name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].values())):
        print(f"{k[0][:100]}   grid {k[1]}")
        for c, x in v.items():
            print(f"    {c:42s} per launch {x / n[(k, c)]:18.1f}   launches {n[(k, c)]}")
