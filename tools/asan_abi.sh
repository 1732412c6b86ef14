#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the C-ABI's argument validation (sanitizers on the CPU side only;
# GPU ASan is not available on this pool). Builds every csrc/*.hip with -fsanitize=address,undefined -fno-gpu-sanitize into a scratch
# library and runs tests/abi_validation.c against it: every call is rejected before any HIP call, so no GPU is needed.
# The build is cached under ${TMPDIR:-/tmp}/ktf_asan, keyed by a checksum of the sources (the CPU suite runs this by default).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/ktf_asan
mkdir -p $OUT
cd $ROOT/kaldi-tflite_amd/csrc
# every source of the product library (csrc/Makefile: SRCS), so that no launcher stays an undefined symbol behind lazy binding
SRCS=$(sed -n 's/^SRCS := //p' Makefile)
KEY=$(cat *.hip *.h *.inc Makefile $ROOT/include/ktf_hip.h $ROOT/tests/abi_validation.c $0 | sha256sum | cut -d' ' -f1)
if [ "$(cat $OUT/key 2>/dev/null)" != "$KEY" ] || [ ! -x $OUT/abi_validation ]; then
    rm -f $OUT/key
    # one compile per source, in parallel (the single-command build took ~90 s)
    pids=""
    for s in $SRCS; do
        /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fsanitize=address,undefined -fno-gpu-sanitize -c $s -o $OUT/${s%.hip}.o &
        pids="$pids $!"
    done
    for p in $pids; do wait $p; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -shared -fPIC $(for s in $SRCS; do echo $OUT/${s%.hip}.o; done) -o $OUT/libktf_asan.so
    /opt/rocm/lib/llvm/bin/clang -g -fsanitize=address,undefined -I$ROOT/include $ROOT/tests/abi_validation.c -o $OUT/abi_validation \
        -L$OUT -lktf_asan -Wl,-rpath,$OUT -Wl,-rpath,/opt/rocm/lib
    echo $KEY > $OUT/key
fi
# (every symbol resolved at load time: a source missing from the build is an error here, not a lazily bound stub nobody calls)
LD_BIND_NOW=1 $OUT/abi_validation
