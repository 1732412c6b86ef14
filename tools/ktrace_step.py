#!/usr/bin/env python3
"""Kernels of the LAST step in a rocprofv3 kernel trace, in launch order with their durations: python tools/ktrace_step.py DIR [first-kernel-substring]
(a step starts at the last launch whose name contains the substring; default: the front-end kernel)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
key = sys.argv[2] if len(sys.argv) > 2 else "frontend"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
step = rows[starts[-2]:starts[-1]] if len(starts) > 1 else rows[starts[-1]:]
t0 = int(step[0]["Start_Timestamp"])
tot = 0
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{d / 1e3:8.1f} us  grid {r.get("Grid_Size_X", "?"):>8s}  {r["Kernel_Name"][:110]}')
print(f"sum of kernel durations {tot / 1e3:.1f} us, span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
