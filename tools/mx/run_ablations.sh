#!/bin/bash
# on the GPU box: per-layer GEMM ms of the product library and of every ablation build present
mkdir -p gpurun_out/abl
for lib in kaldi-tflite_amd/kaldi_tflite_amd/libktf_hip.so kaldi-tflite_amd/kaldi_tflite_amd/libktf_abl_*.so; do
  n=$(basename $lib .so)
  KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=$PWD/$lib python bench.py --gemm f16mx $ABL_ARGS --no-extra --no-cpu-baseline --no-parity 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['per_layer_ms'].items()})" | tee -a gpurun_out/abl/summary.txt
done
