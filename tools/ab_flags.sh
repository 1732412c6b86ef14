#!/bin/bash
# GPU box: headline bench under different bench.py flags: tools/ab_flags.sh <outdir> "<flags A>" "<flags B>" ...
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
for f in "$@"; do
  python3 bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$f]', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['per_layer_ms'].items()}, 'dev', d.get('max_abs_dev_vs_fp64_oracle'))
" | tee -a $OUT/ab.log
done
done
