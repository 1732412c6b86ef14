#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/r6/ holds, from ONE build of the library, into gpurun_out/prof_r6/.
#   kernel trace + stats of the default bench command, the two HBM PMC passes (traffic tied to ktf_build_id), SQ counters of the isolated
#   GEMM layers, the step timeline, the 1.5 s route's per-launch trace, and the full bench line (with other_configs).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r6
rm -rf $OUT; mkdir -p $OUT
BID=$(python3 -c "import sys; sys.path.insert(0, '$REPO/kaldi-tflite_amd'); from kaldi_tflite_amd import ops; print(ops.build_id())")
echo "library build id $BID" | tee $OUT/build_id.txt
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-clock-probe"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $ARGS > $OUT/bench_under_rocprof.log 2>&1
PARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --repeats 1 --no-clock-probe"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o run --output-format csv -- python3 $PARGS > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o run --output-format csv -- python3 $PARGS > $OUT/write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/short -o short --output-format csv -- python3 $REPO/tools/short_profile.py > $OUT/short.log 2>&1
cd $REPO
python3 tools/make_kernel_stats.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_f16mx.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-clock-probe (round 6, f16mx, 1024 x 10 s per step, library build $BID)"
sed -i 's/^Kernel names shortened to 90 characters; torch kernels are the synthetic-input generation.$/Kernel names shortened to 90 characters; torch kernels are the synthetic-input generation. Of the 240 front-end launches 17 belong to the steps (2 warm-up + three timed regions of 5) and the rest to the `mfcc` side measurement of the line; `tdnn_f32t_kernel` is the fp32 comparison of the timed batch (`timed_batch_vs_f32`); the five `tdnn_mx_kernel` launches of a step are `<1, 0, true, true>` (tdnn1), 3 x `<1, 0, false, true>` (tdnn2-4) and `<1, 2, false, true>` (tdnn5 + pooling), all on flat row tiles./' $OUT/kernel_stats_f16mx.md
python3 tools/make_traffic_r3.py $OUT/fetch $OUT/write tdnn_mx $OUT/traffic_f16mx.json 13181000000 $BID > $OUT/traffic.log 2>&1
python3 tools/step_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) --md > $OUT/step_timeline_f16mx.md 2>&1
{ echo "# 1024 x 1.5 s windows, shipped f16mx model (route: split-bf16 on flat row tiles): last step of tools/short_profile.py under rocprofv3 --kernel-trace (build $BID)"; echo; echo '```'; tail -1 $OUT/short.log; python3 tools/ktrace_step.py $OUT/short; echo '```'; } > $OUT/short_route_trace.md 2>&1
grep "^{" $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof_f16mx.json
bash tools/pmc_step.sh r6 f16mx > /dev/null 2>&1; cp $REPO/gpurun_out/pmc_r6/summary.txt $OUT/pmc_f16mx.txt
mkdir -p $REPO/profiles/r6 && cp $OUT/traffic_f16mx.json $REPO/profiles/r6/traffic_f16mx.json      # (this box's copy of the tree: bench.py attaches the traffic of THIS build)
timeout 900 python3 bench.py > $OUT/bench_full.log 2>&1
grep "^{" $OUT/bench_full.log | tail -1 > $OUT/bench_f16mx_final.json
python3 tools/show_bench.py $OUT/bench_f16mx_final.json 2>/dev/null | head -40
rm -rf $OUT/fetch/*/*.db $OUT/write/*/*.db 2>/dev/null
du -sh $OUT
