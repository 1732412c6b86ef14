#!/usr/bin/env python3
"""Formats a rocprofv3 --kernel-trace --stats kernel_stats.csv as the markdown table kept under profiles/.

usage: make_kernel_stats.py <kernel_stats.csv> <out.md> "<title line>"
"""
import csv
import sys

src, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(src)))
with open(out, "w") as f:
    f.write(f"# {title}\n\n")
    f.write("Kernel names shortened to 90 characters; torch kernels are the synthetic-input generation.\n\n")
    f.write("| kernel | calls | total ns | average ns | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {float(r['Percentage']):.2f} |\n")
print(open(out).read())
