#!/bin/bash
# on the GPU box: per-layer GEMM ms of the product library (classic and persistent kernels) and of every ablation build present
# (p_* builds: the persistent kernel; the others: the one-tile-per-workgroup kernel)
mkdir -p gpurun_out/abl
: > gpurun_out/abl/summary.txt
run() {  # name lib args
  KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=$PWD/$2 python bench.py --gemm f16mx $3 --no-extra --no-cpu-baseline --no-parity 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['roofline']['gemm_ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['per_layer_ms'].items()})" | tee -a gpurun_out/abl/summary.txt
}
P=kaldi-tflite_amd/kaldi_tflite_amd
for round in 1 2; do
  run classic $P/libktf_hip.so --mx-classic
  run persist $P/libktf_hip.so --mx-persist
  for lib in $P/libktf_abl_*.so; do
    n=$(basename $lib .so); n=${n#libktf_abl_}
    case $n in p_*) run $n $lib --mx-persist;; *) run $n $lib --mx-classic;; esac
  done
done
