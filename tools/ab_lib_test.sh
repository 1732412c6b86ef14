#!/bin/bash
# GPU box: run the f16x2 / bf16x3 GPU parity tests against a library variant, then the A/B bench. usage: tools/ab_lib_test.sh <outdir> <variant> <baseline>
OUT=$1; V=$2; B=$3
D=$(pwd)/kaldi-tflite_amd/kaldi_tflite_amd
KTF_LIBRARY=$D/libktf_$V.so python3 -m pytest tests -m gpu -x -q -k "f16x2 or full_topology or fused or reproducible or compiled" 2>&1 | tail -3
tools/ab_lib.sh $OUT $B $V
