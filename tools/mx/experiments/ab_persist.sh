#!/bin/bash
# on the GPU box: the persistent f16mx kernel against the one-tile-per-workgroup kernel, alternating runs of bench.py (per-layer GEMM ms)
mkdir -p gpurun_out/abp
: > gpurun_out/abp/summary.txt
for round in 1 2 3; do
  for mode in classic persist; do
    python bench.py --gemm f16mx --mx-$mode --no-extra --no-cpu-baseline --no-parity $ABP_ARGS 2>gpurun_out/abp/err_$mode.txt | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['gemm_ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['per_layer_ms'].items()})" | tee -a gpurun_out/abp/summary.txt
  done
done
