#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory: python tools/kstats.py DIR [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {float(r["Percentage"]):5.1f} %')
