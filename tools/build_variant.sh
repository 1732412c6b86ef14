#!/bin/bash
# Builds a measurement variant of the library: tools/build_variant.sh <name> "<extra hipcc flags>" -> kaldi_tflite_amd/libktf_<name>.so
# (select it with KTF_LIBRARY=<path>; never loaded otherwise)
set -e
NAME=$1; shift
FLAGS="$*"
SRC=$(dirname $0)/../kaldi-tflite_amd/csrc
OUT=$(dirname $0)/../kaldi-tflite_amd/kaldi_tflite_amd/libktf_$NAME.so
TMP=$(mktemp -d)
for f in api frontend frontend512 vad_cmvn tdnn_gemm pool_post; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result $FLAGS -c $SRC/$f.hip -o $TMP/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $TMP/*.o -o $OUT
rm -rf $TMP
ls -la $OUT
