#!/bin/bash
# timing build of the persistent f16mx kernel: libktf_abl_mxpprof.so = the product objects with tdnn_mxp.hip compiled -DKTF_MXP_PROF
set -e
cd "$(dirname "$0")/../../kaldi-tflite_amd/csrc"
make -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DKTF_MXP_PROF=${KTF_MXP_PROF_LEVEL:-1} "$@" -c tdnn_mxp.hip -o /tmp/tdnn_mxp_prof.o
objs=$(sed -n 's/^SRCS := //p' Makefile | sed 's/\.hip/.o/g; s/tdnn_mxp\.o//')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/tdnn_mxp_prof.o -o ../kaldi_tflite_amd/libktf_abl_mxpprof.so
ls -la ../kaldi_tflite_amd/libktf_abl_mxpprof.so
