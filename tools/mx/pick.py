"""Print the per-layer GEMM times of bench.py JSON lines on stdin (A/B helper)."""
import json, sys
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    j = json.loads(line)
    r = j["roofline"]
    print(f'{j["value"]:.0f} x-vec/s  step {j["ms_per_step"]:.3f} ms  gemm {r["gemm_ms_per_step"]:.3f} ms  clk {(j.get("shader_clock_mhz") or 0):.0f}  '
          + "  ".join(f"{k} {v:.3f}" for k, v in r["per_layer_ms"].items()) + f'  mfcc {j["mfcc"]["ms"]:.3f}')
