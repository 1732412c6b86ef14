#!/bin/bash
# on the GPU box: rocprofv3 kernel statistics of the f16mx step for each given library build (kernel durations, not event brackets)
REPO=$(pwd)
mkdir -p $REPO/gpurun_out/ktrace
cd /tmp && export TMPDIR=/tmp
export KTF_ALLOW_LIBRARY_OVERRIDE=1
for lib in "$@"; do
  n=$(basename $lib .so)
  export KTF_LIBRARY=$REPO/$lib
  rm -rf /tmp/kt_$n
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n -o run --output-format csv -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-clock-probe --no-parity --repeats 1 > /tmp/kt_$n.log 2>&1
  echo "== $n" | tee -a $REPO/gpurun_out/ktrace/summary.txt
  python3 $REPO/tools/kstats.py /tmp/kt_$n 8 | tee -a $REPO/gpurun_out/ktrace/summary.txt
done
