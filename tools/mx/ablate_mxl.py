#!/usr/bin/env python3
"""Timing-only ablation builds of csrc/tdnn_mxl.hip (measurement tool; results of these builds are WRONG by design, except `prof`).
As tools/mx/ablate.py: the product source carries no switches; this script patches a scratch copy and builds
libktf_abl_<name>.so beside the product library (run on the GPU box by tools/mx/run_ablations.sh).

    python tools/mx/ablate_mxl.py [variant ...]      (no arguments: all of them)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
src = open(os.path.join(CS, "tdnn_mxl.hip")).read()



def _product_objects():
    """The object files of libktf_hip.so (csrc/Makefile: SRCS), built by `make` beforehand."""
    import re
    srcs = re.search(r"^SRCS := (.*)$", open(os.path.join(CS, "Makefile")).read(), re.M).group(1).split()
    return [f[:-4] + ".o" for f in srcs]

def rep(s, a, b, count=1):
    assert s.count(a) >= 1, a
    return s.replace(a, b) if count == 0 else s.replace(a, b, count)


V = {}
# the loader never waits for its DMAs (matrix waves read whatever is there): issue rate + MFMA + barriers
V["l_nowait"] = lambda s: rep(s, '#define XL_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")', "#define XL_WAIT_VM(n_) ((void)0)")
# no DMA at all: MFMA + fragment reads + barriers
V["l_nodma"] = lambda s: rep(s, '#include "tdnn_mx_common.h"', '#include "tdnn_mx_common.h"\n#define __builtin_amdgcn_global_load_lds(...) ((void)0)')
# no MFMA, no fragment reads: the DMA stream + barriers
V["l_nomfma"] = lambda s: rep(s, "            if (i < nblk) {                                                                                            \\", "            if (false) {                                                                                               \\", 0)
# no epilogue
V["l_noepi"] = lambda s: rep(s, "    const float* prm = reinterpret_cast<const float*>(rsm + XL_PRM_OFF);", "    return;\n    const float* prm = reinterpret_cast<const float*>(rsm + XL_PRM_OFF);")
# side A and side W not fetched (half stages only)
V["l_nosides"] = lambda s: rep(rep(s, "#define XL_SA(buf_, j_)                                                                                                \\\n        {",
                                   "#define XL_SA(buf_, j_)                                                                                                \\\n        if (false) {"),
                               "                if (pc_ < 22) {", "                if (false) {")


# instrumented build (correct results): shader-clock stamps summed per workgroup role and phase part, per (output kind, super-steps)
def _prof(s):
    s = rep(s, "static_assert(XL_LDS_BYTES <= 163840, \"LDS budget\");",
            "static_assert(XL_LDS_BYTES <= 163840, \"LDS budget\");\n__device__ unsigned long long g_xprof[48][16];\n"
            "#define XP_T() ((long long)__builtin_readcyclecounter())")
    # loader: issue time, wait time, barrier time
    s = rep(s, "        const int l = wave - 8;", "        const int l = wave - 8;\n        long long xp_i = 0, xp_w = 0, xp_b = 0, xp_t = XP_T(), xp_pro = 0; const long long xp_0 = xp_t;")
    s = rep(s, '#define XL_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")',
            '#define XL_WAIT_VM(n_) { if (wave >= 8) { const long long t_ = XP_T(); xp_i += t_ - xp_t; xp_t = t_; } asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory"); if (wave >= 8) { const long long t_ = XP_T(); xp_w += t_ - xp_t; xp_t = t_; } }')
    s = rep(s, "#define XL_BARRIER()                                                                                                   \\\n    {                                                                                                                  \\\n        __builtin_amdgcn_s_barrier();                                                                                  \\\n        asm volatile(\"\" ::: \"memory\");                                                                                 \\\n    }",
            "#define XL_BARRIER()                                                                                                   \\\n    {                                                                                                                  \\\n        const long long tb_ = XP_T();                                                                                  \\\n        __builtin_amdgcn_s_barrier();                                                                                  \\\n        asm volatile(\"\" ::: \"memory\");                                                                                 \\\n        { const long long t_ = XP_T(); xp_b += t_ - tb_; xp_t = t_; }                                                  \\\n    }")
    s = rep(s, "        XL_BARRIER()                                 // opens F0 of super-step 0", "        XL_BARRIER()                                 // opens F0 of super-step 0\n        xp_pro = XP_T() - xp_0; xp_i = 0; xp_w = 0; xp_b = 0;")
    s = rep(s, "#undef XL_CTX\n        return;", "#undef XL_CTX\n        if (l == 0 && lane == 0) { unsigned long long* g = g_xprof[(p.nss < 15 ? p.nss : 15) + 16 * OUT];\n"
            "            atomicAdd(g + 0, (unsigned long long)xp_pro); atomicAdd(g + 1, (unsigned long long)xp_i); atomicAdd(g + 2, (unsigned long long)xp_w); atomicAdd(g + 3, (unsigned long long)xp_b); atomicAdd(g + 4, 1ull); }\n        return;")
    # matrix: compute time (barrier exit -> next barrier arrival), barrier time, epilogue
    s = rep(s, "    const int wm = wave >> 2, wn = wave & 3;", "    const int wm = wave >> 2, wn = wave & 3;\n    long long xp_b = 0, xp_t = XP_T(); const long long xp_0 = xp_t; long long xp_w = 0, xp_i = 0; (void)xp_w; (void)xp_i;")
    s = rep(s, "    __builtin_amdgcn_sched_barrier(0);\n#undef XL_F", "    __builtin_amdgcn_sched_barrier(0);\n    const long long xp_k = XP_T();\n#undef XL_F")
    tail = ("{ const long long xp_s = XP_T(); asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); const long long xp_e = XP_T(); if (wave == 0 && lane == 0) { unsigned long long* g = g_xprof[(p.nss < 15 ? p.nss : 15) + 16 * OUT];\n"
            "            atomicAdd(g + 8, (unsigned long long)(xp_k - xp_0)); atomicAdd(g + 9, (unsigned long long)xp_b); atomicAdd(g + 10, (unsigned long long)(xp_s - xp_k)); atomicAdd(g + 12, (unsigned long long)(xp_e - xp_s)); atomicAdd(g + 11, 1ull); } }")
    s = rep(s, "            }\n        }\n        return;\n    } else {", "            }\n        }\n        " + tail + "\n        return;\n    } else {")
    s = rep(s, "                    }\n                }\n            }\n        }\n    }\n}\n\nint mxl_launch", "                    }\n                }\n            }\n        }\n        " + tail + "\n    }\n}\n\nint mxl_launch")
    s += """
extern "C" void ktf_xprof_dump(void) {
    unsigned long long h[48][16];
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_xprof), sizeof(h));
    for (int i = 0; i < 48; ++i)
        if (h[i][4] || h[i][11]) {
            const double nl = (double)(h[i][4] ? h[i][4] : 1), nm = (double)(h[i][11] ? h[i][11] : 1);
            printf("out %d nss %2d tiles %llu | loader clk/tile: prologue %.0f issue %.0f wait %.0f barrier %.0f | matrix clk/tile: k-loop %.0f (barriers %.0f) epilogue %.0f + store acks %.0f\\n",
                   i / 16, i % 16, h[i][11], h[i][0] / nl, h[i][1] / nl, h[i][2] / nl, h[i][3] / nl, h[i][8] / nm, h[i][9] / nm, h[i][10] / nm, h[i][12] / nm);
        }
    hipMemset(0, 0, 0);
    unsigned long long z[48][16] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(g_xprof), z, sizeof(z));
}
"""
    return s


V["l_prof"] = _prof


# instrumented, finer epilogue stamps of the plane epilogue: rows set up | encode loops | wide stores (sums over both chunks)
def _prof_epi(s):
    s = _prof(s)
    s = rep(s, "        const bool affine = p.scale != nullptr;", "        const bool affine = p.scale != nullptr;\n        long long xe_t = XP_T(); const long long xe_rows = xe_t - xp_k; long long xe_enc = 0, xe_st = 0;")
    s = rep(s, "                mx_transpose4(l4r[0], l4r[1], l4r[2], l4r[3]);", "                { const long long t_ = XP_T(); xe_enc += t_ - xe_t; xe_t = t_; }\n                mx_transpose4(l4r[0], l4r[1], l4r[2], l4r[3]);")
    s = rep(s, "            }\n        } else {\n            // fp32 rows (B, T, ldy)",
            "            }\n            { const long long t_ = XP_T(); xe_st += t_ - xe_t; xe_t = t_; }\n"
            "            if (wave == 0 && lane == 0) { unsigned long long* g = g_xprof[(p.nss < 15 ? p.nss : 15) + 16 * OUT]; atomicAdd(g + 13, (unsigned long long)xe_rows); atomicAdd(g + 14, (unsigned long long)xe_enc); atomicAdd(g + 15, (unsigned long long)xe_st); }\n"
            "        } else {\n            // fp32 rows (B, T, ldy)")
    s = rep(s, "h[i][10] / nm, h[i][12] / nm);", "h[i][10] / nm, h[i][12] / nm);\n            if (h[i][13]) printf(\"      plane epilogue: rows %.0f encode loops %.0f transposes + wide stores %.0f\\n\", h[i][13] / nm, h[i][14] / nm, h[i][15] / nm);")
    return s


V["l_prof_epi"] = _prof_epi
# A/B (correct results): static wave priorities
V["l_prio_matrix"] = lambda s: rep(s, "    const int wm = wave >> 2, wn = wave & 3;", "    __builtin_amdgcn_s_setprio(3);\n    const int wm = wave >> 2, wn = wave & 3;")
V["l_prio_loader"] = lambda s: rep(s, "        const int l = wave - 8;", "        __builtin_amdgcn_s_setprio(3);\n        const int l = wave - 8;")
V["l_prio_young"] = lambda s: rep(s, "    const int wm = wave >> 2, wn = wave & 3;", "    if (wave >= 4) __builtin_amdgcn_s_setprio(1);\n    const int wm = wave >> 2, wn = wave & 3;")
# A/B (correct results, instrumented): how many row blocks the plane encoder's scheduling window spans
for _n, _c in ():
    V["l_prof_epi_sb" + _n] = (lambda c: lambda s: _prof_epi(rep(s, "if (i & 1) __builtin_amdgcn_sched_barrier(0);", "if (" + c + ") __builtin_amdgcn_sched_barrier(0);")))(_c)






if __name__ == "__main__":
    names = sys.argv[1:] or list(V)
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result -Wno-unused-value".split()
    objs = [os.path.join(CS, o) for o in _product_objects() if o != "tdnn_mxl.o"]
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
