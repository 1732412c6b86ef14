#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...):
the launches of the LAST complete step (front-end launch to front-end launch), their durations and the idle gaps between them.
   python tools/step_timeline.py gpurun_out/prof/NAME_results.db | NAME_kernel_trace.csv [--md]"""
import sys

if sys.argv[1].endswith(".csv"):          # rocprofv3 --kernel-trace --output-format csv: <name>_kernel_trace.csv
    import csv
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))),
                  key=lambda r: r[1])
else:
    import sqlite3
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
# front-end launches of the full-size (timed) steps: the parity pass behind them runs a handful of utterances (tens of microseconds)
longest = max((r[2] - r[1] for r in rows if "frontend" in r[0]), default=0)
fe = [i for i, r in enumerate(rows) if "frontend" in r[0] and (r[2] - r[1]) > 0.5 * longest]
# two consecutive full-size front-end launches less than 100 ms apart bracket one timed step (later full-size launches belong to
# the fp32 comparison of the timed batch): the last such pair
starts = [i for i in fe if i + 1 < len(rows) and "frontend" not in rows[i + 1][0]]
pairs = [(x, y) for x, y in zip(starts, starts[1:]) if rows[y][1] - rows[x][1] < 100e6]
a, b = pairs[-1]
seg = rows[a:b]
span = rows[b][1] - rows[a][1]
md = "--md" in sys.argv
if md:
    print("| kernel | start us | duration us |\n|---|---:|---:|")
busy = 0
for n, s, e in seg:
    busy += e - s
    name = n.split("(")[0].replace("void ", "")
    print((f"| `{name}` | {(s - seg[0][1]) / 1e3:.1f} | {(e - s) / 1e3:.1f} |") if md else f"{(s - seg[0][1]) / 1e3:10.1f} {(e - s) / 1e3:9.1f} us  {name}")
print(("\n" if md else "") + f"step {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
