#!/bin/bash
# GPU box: headline bench for a list of library variants (tools/build_variant.sh): tools/ab_lib.sh <outdir> <name>...
OUT=$1; shift
mkdir -p $OUT
D=$(pwd)/kaldi-tflite_amd/kaldi_tflite_amd
for rep in 1 2; do
for v in "$@"; do
  KTF_LIBRARY=$D/libktf_$v.so python3 bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['per_layer_ms'].items()}, 'mfcc', round(d['mfcc']['ms'],3), 'dev', d.get('max_abs_dev_vs_fp64_oracle'))
" | tee -a $OUT/ab.log
done
done
