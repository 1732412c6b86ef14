#!/usr/bin/env python3
"""Timing-only ablation builds of csrc/tdnn_mxs.hip (measurement tool; results of these builds are WRONG by design). As
tools/mx/ablate.py: the product source carries no switches; this script patches a scratch copy and builds libktf_abl_<name>.so beside
the product library (run on the GPU box by `ABL_ARGS=--mx-slab tools/mx/run_ablations.sh`).

    python tools/mx/ablate_mxs.py [variant ...]      (no arguments: all of them)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
src = open(os.path.join(CS, "tdnn_mxs.hip")).read()



def _product_objects():
    """The object files of libktf_hip.so (csrc/Makefile: SRCS), built by `make` beforehand."""
    import re
    srcs = re.search(r"^SRCS := (.*)$", open(os.path.join(CS, "Makefile")).read(), re.M).group(1).split()
    return [f[:-4] + ".o" for f in srcs]

def rep(s, a, b, count=1):
    if a == "":
        return s
    assert s.count(a) >= 1, a
    return s.replace(a, b) if count == 0 else s.replace(a, b, count)


V = {}
V["s_asbuilt"] = lambda s: s
# no DMA at all: MFMA + fragment reads + barriers
V["s_nodma"] = lambda s: rep(s, '#include "tdnn_mx_common.h"', '#include "tdnn_mx_common.h"\n#define __builtin_amdgcn_global_load_lds(...) ((void)0)')
# the slabs are not fetched (W stages and the W side only)
V["s_noslab"] = lambda s: rep(s, "                if (slab_next) {", "                if (false) {")
# the W stages are not fetched
V["s_now"] = lambda s: rep(s, "                if (next) {\n                    if (i == 0) XS_DMA_W", "                if (false) {\n                    if (i == 0) XS_DMA_W")
# the W side is not fetched
V["s_nosw"] = lambda s: rep(s, "                if (j == 1) {                        // (issued LAST", "                if (false) {                        // (issued LAST")
# nobody waits for a DMA to land
V["s_nowait"] = lambda s: rep(rep(s, 'if (j == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");', "if (false) {}"), 'else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', "")
# every tile reads the slabs of utterance 0, chunk 0 (L2 hits): what the COLD activation fetches cost
V["s_hot"] = lambda s: rep(rep(s, "const int64_t ub = (int64_t)b * p.nch_in * p.T;", "const int64_t ub = 0;"), "const char* g_ = sl_ptr[s_] + (unsigned)(c_) * sl_cs[s_];", "const char* g_ = sl_ptr[s_];")
# ... of the tile's own utterance but always chunk 0 (cold once per tile)
V["s_hot_chunk"] = lambda s: rep(s, "const char* g_ = sl_ptr[s_] + (unsigned)(c_) * sl_cs[s_];", "const char* g_ = sl_ptr[s_];")
# all four slots of a slab go out in the chunk's first K-step
V["s_burst"] = lambda s: rep(rep(rep(rep(s, "if (k_ci == 1) XS_SLAB(k_c + 1, 1)", "if (k_ci == 0) XS_SLAB(k_c + 1, 1)"), "if (k_ci == s2_ci) XS_SLAB(k_c + 1, 2)", "if (k_ci == 0) XS_SLAB(k_c + 1, 2)"),
                                 "if (k_ci == s3_ci) XS_SLAB(k_c + 1, 3)", "if (k_ci == 0) XS_SLAB(k_c + 1, 3)"), "", "")
# ... in its last K-step
V["s_burst_last"] = lambda s: rep(rep(rep(rep(s, "if (k_ci == 1) XS_SLAB(k_c + 1, 1)", "if (k_ci == p.nctx - 1) XS_SLAB(k_c + 1, 1)"), "if (k_ci == s2_ci) XS_SLAB(k_c + 1, 2)", "if (k_ci == p.nctx - 1) XS_SLAB(k_c + 1, 2)"),
                                 "if (k_ci == s3_ci) XS_SLAB(k_c + 1, 3)", "if (k_ci == p.nctx - 1) XS_SLAB(k_c + 1, 3)"), "if (k_ci == 0) XS_SLAB(k_c + 1, 0)", "if (k_ci == p.nctx - 1) XS_SLAB(k_c + 1, 0)")

if __name__ == "__main__":
    names = sys.argv[1:] or list(V)
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result -Wno-unused-value".split()
    objs = [os.path.join(CS, o) for o in _product_objects() if o != "tdnn_mxs.o"]
    for n in names:
        scratch = os.path.join(CS, f"_abl_{n}.hip")
        open(scratch, "w").write(V[n](src))
        obj = os.path.join(CS, f"_abl_{n}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", scratch, "-o", obj])
        out = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", f"libktf_abl_{n}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [obj, "-o", out])
        os.remove(scratch)
        os.remove(obj)
        print("built", out)
