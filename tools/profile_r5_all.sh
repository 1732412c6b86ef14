#!/bin/bash
# GPU box: tools/profile_r5.sh (f16mx) + the two HBM PMC passes of the split-bf16 mode, all from the build in the tree
bash tools/profile_r5.sh > /dev/null 2>&1
cat gpurun_out/prof_r5/build_id.txt; tail -1 gpurun_out/prof_r5/step_timeline_f16mx.md
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r5_bf16x3; rm -rf $OUT; mkdir -p $OUT
BID=$(python3 -c "import sys; sys.path.insert(0, '$REPO/kaldi-tflite_amd'); from kaldi_tflite_amd import ops; print(ops.build_id())")
cd /tmp && export TMPDIR=/tmp
PARGS="$REPO/bench.py --gemm bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --repeats 1 --no-clock-probe"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o run --output-format csv -- python3 $PARGS > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o run --output-format csv -- python3 $PARGS > $OUT/write.log 2>&1
cd $REPO
python3 tools/make_traffic_r3.py $OUT/fetch $OUT/write tdnn_x3 $OUT/traffic_bf16x3.json 17100000000 $BID 2>&1 | grep corrected | head -1
rm -rf $OUT/fetch/*/*.db $OUT/write/*/*.db
