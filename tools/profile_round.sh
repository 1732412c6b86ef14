#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + the two HBM PMC passes of the default bench workload.
# usage: tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra ${BENCH_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $ARGS > $OUT/trace.log 2>&1
PARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS:-}"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o run --output-format csv -- python3 $PARGS > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o run --output-format csv -- python3 $PARGS > $OUT/write.log 2>&1
tail -1 $OUT/trace.log
find $OUT -name "*.csv" | head -20
