#!/bin/bash
# on the GPU box: the f16mx step with each of the given library builds in turn, ROUNDS times (same-box A/B in alternating order);
# per-layer GEMM times by tools/mx/pick.py into gpurun_out/ab.txt.   tools/ab_libs.sh <lib.so> <lib.so> ...   (ROUNDS=3, BENCH_ARGS=...)
mkdir -p gpurun_out
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in "$@"; do
    echo -n "$(basename $lib .so) " | tee -a gpurun_out/ab.txt
    KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=$PWD/$lib python bench.py --gemm f16mx --no-extra --no-cpu-baseline --no-parity --repeats 1 $BENCH_ARGS 2>/dev/null | python tools/mx/pick.py | tee -a gpurun_out/ab.txt
  done
done
