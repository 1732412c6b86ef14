#!/bin/bash
# GPU box: A/B of the split-bf16 kernel choices with the probe library (knobs from the environment). usage: tools/ab_x3.sh <outdir>
OUT=${1:-gpurun_out/ab_x3}
mkdir -p $OUT
export KTF_LIBRARY=$(pwd)/kaldi-tflite_amd/kaldi_tflite_amd/libktf_probe.so
for v in 1 2 1 2; do
  KTF_X3S=$v python3 bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('X3S=$v', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['per_layer_ms'].items()}, 'dev', d.get('max_abs_dev_vs_fp64_oracle'))
" | tee -a $OUT/ab.log
done
