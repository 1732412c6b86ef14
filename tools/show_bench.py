import json
import sys
d=json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3i/bench_default.json"))
print(d["dtype"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["per_layer_ms"])
print(d["max_abs_dev_by_input"], d["tolerance_ok"], d["timed_batch_vs_f32"])
for k,v in d["other_configs"].items():
    if isinstance(v, dict) and "x_vectors_per_s" in v: print(k, round(v["x_vectors_per_s"]), v.get("max_abs_dev_by_input"), v.get("tolerance_ok"), (v.get("roofline") or {}).get("frac"))
    else: print(k, v)
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["legs"])
