#!/usr/bin/env python3
"""profiles/r3/traffic_<mode>.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately) of bench.py:
HBM bytes of the FULL-SIZE TDNN GEMM launches only (grid >= 2^20 threads: the timed 1024-utterance steps; the parity pass and
the fp32 comparison model launch other grids / kernels), per step = per five launches.

usage: make_traffic_r3.py <fetch_dir> <write_dir> <kernel name prefix> <out.json> <algorithmic_gemm_bytes_per_step> [library build id]
(the build id -- ktf_build_id() of the profiled library -- ties the figure to its build: bench.py attaches it only to that build)

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts the 128-byte requests of 16 B/lane
coalesced reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores. Both counters are in KB."""
import collections, csv, glob, json, re, sys


def collect(d, counter, prefix):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or int(r["Grid_Size"]) < (1 << 20):
                continue
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if k.startswith(prefix):
                per[(k, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per


fetch_dir, write_dir, prefix, out, alg = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
F, W = collect(fetch_dir, "FETCH_SIZE", prefix), collect(write_dir, "WRITE_SIZE", prefix)
rows, tot_f, tot_w, n = {}, 0.0, 0.0, 0
for key in sorted(set(F) | set(W)):
    f = sum(F[key]) / max(len(F[key]), 1)
    w = sum(W[key]) / max(len(W[key]), 1)
    rows[f"{key[0]} grid {key[1]}"] = {"launches_seen": len(F[key]), "fetch_KB_per_launch_raw": f, "write_KB_per_launch": w,
                                         "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024}
    tot_f += sum(F[key]); tot_w += sum(W[key]); n += len(F[key])
steps = n / 5.0
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py, MI355X, B = 1024 x 10 s, full-size TDNN GEMM launches only",
       "units": "KB per counter; gfx950 correction: FETCH_SIZE x 2, WRITE_SIZE exact", "full_size_launches": n, "steps": steps, "kernels": rows,
       "tdnn_gemm_bytes_per_step_corrected": (2 * tot_f + tot_w) * 1024 / steps, "tdnn_gemm_algorithmic_bytes_per_step": alg}
if len(sys.argv) > 6:
    doc["library_build_id"] = sys.argv[6]
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k != "kernels"}, indent=1))
for k, v in rows.items():
    print(k, v)
