#!/bin/bash
# Polls the shader clock while a workload runs: tools/clock_watch.sh <python script + args...>
( for i in $(seq 1 ${POLLS:-12}); do sleep 0.5; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; done ) &
W=$!
python3 "$@" > /dev/null 2>&1
wait $W
