#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...):
the launches of the LAST complete step (front-end launch to front-end launch), their durations and the idle gaps between them.
   python tools/step_timeline.py gpurun_out/prof/NAME_results.db [--md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
fe = [i for i, r in enumerate(rows) if "frontend" in r[0]]
# the last two front-end launches that are followed by other kernels bracket one whole step
starts = [i for i in fe if i + 1 < len(rows) and "frontend" not in rows[i + 1][0]]
a, b = starts[-2], starts[-1]
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
