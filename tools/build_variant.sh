#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> <source.hip> [extra hipcc flags]  ->  _ab/libktf_<name>.so
# (the product objects of csrc/ with <source.hip> recompiled under the extra flags; REPLACES=tdnn_mx.hip: <source.hip> is a patched copy that stands in for it)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/kaldi-tflite_amd/csrc
name=$1; src=$2; shift 2
make -C $CS -j6 >/dev/null
mkdir -p $ROOT/_ab /tmp/ktf_variant
obj=/tmp/ktf_variant/${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c $CS/$src -o $obj
objs=""
for s in $(grep '^SRCS :=' $CS/Makefile | cut -d= -f2); do
  o=$CS/${s%.hip}.o
  [ "$s" == "${REPLACES:-$src}" ] && o=$obj
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $ROOT/_ab/libktf_$name.so
echo built $ROOT/_ab/libktf_$name.so
