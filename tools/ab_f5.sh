#!/bin/bash
# GPU box: front-end kernel variants (tools/build_variant.sh f5v*): goldens + time of 1024 x 998 frames
OUT=${1:-gpurun_out/ab_f5}
mkdir -p $OUT
D=$(pwd)/kaldi-tflite_amd/kaldi_tflite_amd
for v in hip f5v1 f5v2 f5v3 f5v4; do
  export KTF_LIBRARY=$D/libktf_$v.so
  echo "== $v" | tee -a $OUT/ab.log
  python3 tools/microbench.py frontend --iters 20 2>/dev/null | tee -a $OUT/ab.log
  python3 tools/microbench.py frontend --iters 20 2>/dev/null | tee -a $OUT/ab.log
done
