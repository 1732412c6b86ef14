#!/bin/bash
# Polls the shader clock while a workload runs: tools/clock_watch.sh <python args...>
( for i in $(seq 1 12); do sleep 0.5; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; done ) &
W=$!
N=${N:-20000} python3 "$@" > /dev/null 2>&1
wait $W
