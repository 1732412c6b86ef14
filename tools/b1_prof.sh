#!/bin/bash
# Runs on the GPU box: kernel durations of single-utterance calls (tools/batch1_trace.py under rocprofv3 --kernel-trace --stats).
# usage: tools/b1_prof.sh <tag> [gemm]    -> gpurun_out/b1_<tag>/
set -u
TAG=${1:-b1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/b1_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=200 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o run --output-format csv -- python3 $REPO/tools/batch1_trace.py ${2:-f32} > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    calls = int(r["Calls"])
    if calls >= 150:
        per_call = float(r["TotalDurationNs"]) / 200 / 1e3
        tot += per_call
        print(f"{per_call:8.2f} us/utt  calls/utt {calls / 200:4.1f}  avg {float(r['AverageNs']) / 1e3:7.2f}  {r['Name'][:110]}")
print(f"{tot:8.2f} us of kernels per utterance")
PY
