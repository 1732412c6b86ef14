"""Condensed view of one bench.py JSON line: python tools/show_bench.py [path]"""
import json
import sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3i/bench_default.json"))
print(d["dtype"], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4), {k: round(v, 3) for k, v in d["roofline"]["per_layer_ms"].items()})
if "max_abs_dev_by_input" in d:
    print(d["max_abs_dev_by_input"], d["tolerance_ok"], d.get("timed_batch_vs_f32"))
for k, v in (d.get("other_configs") or {}).items():
    if isinstance(v, dict) and "x_vectors_per_s" in v:
        print(k, round(v["x_vectors_per_s"]), v.get("max_abs_dev_by_input"), v.get("tolerance_ok"), (v.get("roofline") or {}).get("frac"))
    else:
        print(k, v)
if d.get("cpu_baseline"):
    print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("legs"))
