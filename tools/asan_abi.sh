#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the C-ABI's argument validation (sanitizers on the CPU side only;
# GPU ASan is not available on this pool). Builds every csrc/*.hip with -fsanitize=address,undefined -fno-gpu-sanitize into a scratch
# library and runs tests/abi_validation.c against it: every call is rejected before any HIP call, so no GPU is needed.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/ktf_asan
mkdir -p $OUT
cd $ROOT/kaldi-tflite_amd/csrc
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fsanitize=address,undefined -fno-gpu-sanitize -shared \
    api.hip tdnn_gemm.hip tdnn_mx.hip pool_post.hip tdnn_f32.hip tdnn_bf16.hip tdnn_split.hip frontend.hip frontend512.hip vad_cmvn.hip -o $OUT/libktf_asan.so
/opt/rocm/lib/llvm/bin/clang -g -fsanitize=address,undefined -I$ROOT/include $ROOT/tests/abi_validation.c -o $OUT/abi_validation \
    -L$OUT -lktf_asan -Wl,-rpath,$OUT -Wl,-rpath,/opt/rocm/lib
$OUT/abi_validation
