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

# optional 4th argument: the kernel_trace.csv of the same run -> a second table without each kernel's first launch
if len(sys.argv) > 4:
    import collections
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[4])):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    slow_first = [k for k, v in d.items() if len(v) > 2 and v[0] > 5 * sorted(v[1:])[len(v[1:]) // 2]]
    with open(out, "a") as f:
        f.write("\n## Steady state (same run, kernel_trace.csv, first launch of every kernel dropped)\n\n")
        if slow_first:
            f.write("First launches more than 5x slower than the kernel's median in this run (one-off, inside the warm-up): "
                    + ", ".join(f"`{k[:50]}`" for k in slow_first) + "; the table above averages them in, this one does not.\n\n")
        f.write("| kernel | calls | average ns | min ns | max ns |\n|---|---:|---:|---:|---:|\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1][1:])):
            v = v[1:]
            if v and not k.startswith("void at::") and not k.startswith("__amd"):
                f.write(f"| `{k[:90]}` | {len(v)} | {sum(v) / len(v):.0f} | {min(v)} | {max(v)} |\n")
    # launches of one kernel with different grids are different populations (the 1024-utterance step, the four-utterance
    # parity pass): the averages above mix them, this table does not -- it is the one bench.py's avg_launch_ms agrees with
    g = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(sys.argv[4])):
        if r["Kernel_Name"] not in seen:          # a kernel's first launch in the process (code load, cold caches: 10-30 ms) is not a sample
            seen.add(r["Kernel_Name"])
            continue
        g[(r["Kernel_Name"], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "a") as f:
        f.write("\n## Per (kernel, grid size) — the TDNN GEMM launches of the timed step vs the parity pass\n\n")
        f.write("| kernel | grid (threads) | calls | average ns | min ns | max ns |\n|---|---:|---:|---:|---:|---:|\n")
        tot_ns, tot_n = 0, 0
        for (k, gs), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
            if k.startswith("void tdnn_") or k.startswith("tdnn_"):
                f.write(f"| `{k[:90]}` | {gs} | {len(v)} | {sum(v) / len(v):.0f} | {min(v)} | {max(v)} |\n")
                if int(gs) >= (1 << 20):
                    tot_ns += sum(v)
                    tot_n += len(v)
        if tot_n:
            f.write(f"\nMean duration of the {tot_n} full-size TDNN GEMM launches (grid >= 2^20 threads): **{tot_ns / tot_n / 1e6:.3f} ms** "
                    "(each kernel's first launch dropped; `roofline.avg_launch_ms` of the bench line is the same quantity measured with HIP events "
                    "over the timed steps).\n")
    print(open(out).read()[-2500:])
