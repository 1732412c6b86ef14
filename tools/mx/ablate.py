#!/usr/bin/env python3
"""ROUNDS 3-5: its text patches match csrc/tdnn_mx.hip as of commit 1d9b8c1 (`git show 1d9b8c1:kaldi-tflite_amd/csrc/tdnn_mx.hip`); round 6
rewrote the K-loop and ablates it with tools/mx/ablate6.py (same idea, new anchors).
Timing-only ablation builds of csrc/tdnn_mx.hip (measurement tool; results of these builds are WRONG by design). The product
source carries no ablation switches: this script patches a scratch copy, builds libktf_abl_<name>.so beside the product
library and prints the `KTF_LIBRARY=... KTF_ALLOW_LIBRARY_OVERRIDE=1 python bench.py --gemm f16mx --no-extra --no-cpu-baseline`
lines to run on the GPU box (tools/mx/run_ablations.sh does)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "kaldi-tflite_amd", "csrc")
src = open(os.path.join(CS, "tdnn_mx.hip")).read()


def _product_objects():
    """The object files of libktf_hip.so (csrc/Makefile: SRCS), built by `make` beforehand."""
    import re
    srcs = re.search(r"^SRCS := (.*)$", open(os.path.join(CS, "Makefile")).read(), re.M).group(1).split()
    return [f[:-4] + ".o" for f in srcs]

def rep(s, a, b, count=1):
    assert s.count(a) >= 1, a
    return s.replace(a, b) if count == 0 else s.replace(a, b, count)

V = {}
V["no_encode"] = lambda s: rep(rep(rep(rep(s,
    "                        mx_encode32(v, hp, l4, h4, sw_);",
    "                        for (int k = 0; k < 4; ++k) hp[k] = u32x4{__float_as_uint(v[8*k]), __float_as_uint(v[8*k+2]), __float_as_uint(v[8*k+4]), __float_as_uint(v[8*k+6])}; l4 = hp[0]; h4 = hp[1]; sw_ = 0;"),
    "                        __builtin_nontemporal_store(l4, reinterpret_cast<u32x4*>(p.yl4 + rec * 16));", "(void)rec;"),
    "                        __builtin_nontemporal_store(h4, reinterpret_cast<u32x4*>(p.y4 + rec * 16));", ""),
    "                        __builtin_nontemporal_store(sw_, reinterpret_cast<unsigned*>(p.ys + rec * 4));", "")
V["no_mx_mfma"] = lambda s: rep(s, "            for (int i = 0; i < 8; ++i) {\n                const u32x4 l = l_n, h = h_n;", "            for (int i = 0; i < 0; ++i) {\n                const u32x4 l = l_n, h = h_n;")
V["no_side_dma"] = lambda s: rep(rep(s, "                if (j == 0) {\n                    if (i == 3) MX_SA_SETUP", "                if (false) {\n                    if (i == 3) MX_SA_SETUP"),
                                 "                if (j == 1) {\n                    if (i == 4)", "                if (false) {\n                    if (i == 4)")
V["no_f16_dma"] = lambda s: rep(s, "                if (next) {\n                    if (i == 0) MX_DMA_F16", "                if (false) {\n                    if (i == 0) MX_DMA_F16")
V["no_dma"] = lambda s: V["no_f16_dma"](V["no_side_dma"](s))
V["no_mfma"] = lambda s: rep(V["no_mx_mfma"](s), "                if (live) {\n                    hfrag8 a_nxt", "                if (false) {\n                    hfrag8 a_nxt")
V["no_epilogue"] = lambda s: rep(s, "        for (int pass = 0; pass < 2; ++pass) {           // rows wm*128", "        for (int pass = 0; pass < 0; ++pass) {           // rows wm*128")

V["hot_a"] = lambda s: rep(rep(rep(s, "const int64_t ub = (int64_t)b * p.nch_in * p.T;", "const int64_t ub = 0;"),
                                "        a_row[i] = t0 + row;", "        a_row[i] = row;"),
                            "        int r_ = t0 + rg_ * 64 + lane + off__;", "        int r_ = rg_ * 64 + lane + off__;")
# ... only the workgroups of the N-tiles behind the first read the hot region (the first N-tile of every M-tile still touches its own
# rows): how much of the cold-fetch cost do the OTHER N-tiles of an M-tile pay today? (they start together with the first one)
V["hot_a_nt"] = lambda s: rep(rep(rep(s, "const int64_t ub = (int64_t)b * p.nch_in * p.T;", "const int64_t ub = nt ? 0 : (int64_t)b * p.nch_in * p.T;"),
                                   "        a_row[i] = t0 + row;", "        a_row[i] = (nt ? 0 : t0) + row;"),
                               "        int r_ = t0 + rg_ * 64 + lane + off__;", "        int r_ = (nt ? 0 : t0) + rg_ * 64 + lane + off__;")
def _no_stores(s):
    import re
    s = re.sub(r"__builtin_nontemporal_store\(([^,]+), (reinterpret_cast<[^;]+)\);", r"if (abl_word(\1) == 0x12345678u) __builtin_nontemporal_store(\1, \2);", s)
    return rep(s, '#include "tdnn_mx_common.h"', '#include "tdnn_mx_common.h"\nstatic __device__ __forceinline__ unsigned abl_word(unsigned v) { return v; }\n'
               "template <typename V> static __device__ __forceinline__ unsigned abl_word(V v) { return v.x; }")
V["no_stores"] = _no_stores

V["prio_hi_half"] = lambda s: rep(s, "    for (int ss = 0; ss < p.nss; ++ss) {", "    if (wave >= 4) __builtin_amdgcn_s_setprio(1);\n    for (int ss = 0; ss < p.nss; ++ss) {")
V["prio_lo_half"] = lambda s: rep(s, "    for (int ss = 0; ss < p.nss; ++ss) {", "    if (wave < 4) __builtin_amdgcn_s_setprio(1);\n    for (int ss = 0; ss < p.nss; ++ss) {")

# A/B (correct results): the half pieces stored straight from the encoding thread (4 x 16 B at a 64-byte lane stride) instead of
# through the in-place LDS image and coalesced 1 KiB store instructions
V["direct_half_stores"] = lambda s: rep(rep(s,
    "                        for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(src + k * 4) = hp[k];\n                        const int64_t rec = rec0 + m;",
    "                        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(hp[k], reinterpret_cast<u32x4*>(p.yh + (rec0 + m) * 64) + k);\n                        const int64_t rec = rec0 + m;"),
    "                if (chunk_ok) {\n                    const unsigned char* hsrc", "                if (false) {\n                    const unsigned char* hsrc")
# A/B (correct results): ordinary stores instead of nontemporal ones
import re
V["plain_stores"] = lambda s: re.sub(r"__builtin_nontemporal_store\(([^,]+), (reinterpret_cast<[^;]+)\);", r"*(\2) = \1;", s)

# instrumented build (correct results): per-workgroup 100 MHz timestamps summed per (output kind, super-step count); dumped by
# ktf_prof_dump() (tools/mx/prof_phases.py). t_k = start .. end of the K-loop, t_e = epilogue until the last store is issued,
# t_d = until the stores are acknowledged.
def _prof(s):
    s = rep(s, '#include "tdnn_mx_common.h"', '#include "tdnn_mx_common.h"\n__device__ unsigned long long g_prof[48][12];')
    s = rep(s, "    const int n0 = nt * 256, t0 = mt * 256;", "    const int n0 = nt * 256, t0 = mt * 256;\n    const long long pt0 = wall_clock64();\n    const long long pc0 = clock64();\n    long long pe[5] = {0, 0, 0, 0, 0};")
    s = rep(s, "    const int rows_valid = len - t0;", "    const long long pt1 = wall_clock64();\n    const long long pc1 = clock64();\n    const int rows_valid = len - t0;")
    s = rep(s, "            __builtin_amdgcn_s_barrier();\n            asm volatile(\"\" ::: \"memory\");\n            constexpr bool live = true;",
            "            __builtin_amdgcn_s_barrier();\n            asm volatile(\"\" ::: \"memory\");\n            if (ks == 0) ppro = wall_clock64() - pt0;\n            constexpr bool live = true;")
    s = rep(s, "    long long pe[5] = {0, 0, 0, 0, 0};", "    long long pe[5] = {0, 0, 0, 0, 0}, ppro = 0;")
    dump = ("{ const long long pt2 = wall_clock64(); __builtin_amdgcn_s_waitcnt(0); __syncthreads(); const long long pt3 = wall_clock64();"
            " if (tid == 0) { unsigned long long* g = g_prof[(p.nss < 15 ? p.nss : 15) + 16 * OUT]; atomicAdd(g, (unsigned long long)(pt1 - pt0));"
            " atomicAdd(g + 1, (unsigned long long)(pt2 - pt1)); atomicAdd(g + 2, (unsigned long long)(pt3 - pt2)); atomicAdd(g + 3, 1ull);"
            " for (int e = 0; e < 5; ++e) atomicAdd(g + 4 + e, (unsigned long long)pe[e]); atomicAdd(g + 9, (unsigned long long)ppro); atomicAdd(g + 10, (unsigned long long)(pc1 - pc0)); } }")
    s = rep(s, "        }\n        return;\n    } else {", "        }\n        " + dump + "\n        return;\n    } else {")
    s = rep(s, "            __syncthreads();\n        }\n    }\n}\n\n// (The tile body is a function", "            __syncthreads();\n            pe[4] += wall_clock64() - pq;\n        }\n        " + dump + "\n    }\n}\n\n// (The tile body is a function")
    # finer epilogue phases (sums over both passes): skew wait | accumulators -> staging | encode | half stores | closing barrier
    s = rep(s, "        __syncthreads();                                 // every fragment read is done", "        __syncthreads();\n        pe[0] = wall_clock64() - pt1;\n        long long pq = wall_clock64();     // every fragment read is done")
    s = rep(s, "            __syncthreads();\n            if constexpr (OUT == MX_OUT_PLANES) {", "            __syncthreads();\n            pe[1] += wall_clock64() - pq; pq = wall_clock64();\n            if constexpr (OUT == MX_OUT_PLANES) {")
    s = rep(s, "                // the half pieces were written by this wave's own lanes", "                pe[2] += wall_clock64() - pq; pq = wall_clock64();\n                // the half pieces were written by this wave's own lanes")
    s = rep(s, "            } else {\n                const int nl = lane * 4;", "                pe[3] += wall_clock64() - pq; pq = wall_clock64();\n            } else {\n                const int nl = lane * 4;")
    s += """
extern "C" void ktf_prof_dump(void) {
    unsigned long long h[48][12];
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h));
    for (int i = 0; i < 48; ++i)
        if (h[i][3]) {
            const double n = (double)h[i][3] * 100.0;
            printf("out %d nss %2d: %llu tiles  prologue %.2f us  K-loop(+prologue) %.2f us  epilogue %.2f us  store-drain %.2f us | skew %.2f  acc->staging %.2f  encode %.2f  half stores %.2f  closing barrier %.2f | K-loop: %.0f clock64 ticks per 10 ns wall tick x 100 = %.0f MHz\\n",
                   i / 16, i % 16, h[i][3], h[i][9] / n, h[i][0] / n, h[i][1] / n, h[i][2] / n, h[i][4] / n, h[i][5] / n, h[i][6] / n, h[i][7] / n, h[i][8] / n, (double)h[i][10] / (double)h[i][0], 100.0 * (double)h[i][10] / (double)h[i][0]);
        }
    memset(h, 0, sizeof(h));
    hipMemcpyToSymbol(HIP_SYMBOL(g_prof), h, sizeof(h));
}
"""
    return s
V["prof"] = _prof


# instrumented build (correct results), two stamps per tile: the K-loop (accumulator init .. last M) in shader clocks and 100 MHz wall
# ticks, and the rest of the tile (entry .. K-loop, K-loop .. exit) -- the same quantities tools/mx/prof_mxp.py prints for tdnn_mxp.hip
def _prof2(s):
    s = rep(s, '#include "tdnn_mx_common.h"', '#include "tdnn_mx_common.h"\n__device__ unsigned long long g_prof2[48][2][8];')
    s = rep(s, "    const int n0 = nt * 256, t0 = mt * 256;", "    const int n0 = nt * 256, t0 = mt * 256;\n    const unsigned long long pc0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();")
    s = rep(s, "    f32x4 acc[8][4];\n", "    const unsigned long long pc1 = __builtin_amdgcn_s_memtime(), pr1 = __builtin_amdgcn_s_memrealtime();\n    f32x4 acc[8][4];\n")
    s = rep(s, "#undef MX_DMA_F16\n", "    const unsigned long long pc2 = __builtin_amdgcn_s_memtime(), pr2 = __builtin_amdgcn_s_memrealtime();\n#undef MX_DMA_F16\n")
    dump = ("{ const unsigned long long pc3 = __builtin_amdgcn_s_memtime(), pr3 = __builtin_amdgcn_s_memrealtime();"
            " const int pw = wave == 0 ? 0 : (wave == 4 ? 1 : -1);"
            " if (pw >= 0 && lane == 0) { unsigned long long* g = g_prof2[(p.nss < 15 ? p.nss : 15) + 16 * OUT][pw]; atomicAdd(g + 3, pc2 - pc1); atomicAdd(g + 4, pr2 - pr1);"
            " atomicAdd(g + 6, (pc1 - pc0) + (pc3 - pc2)); atomicAdd(g + 5, (pr1 - pr0) + (pr3 - pr2)); atomicAdd(g + 7, 1ull); } }")
    inc = open(os.path.join(CS, "tdnn_mx_epilogue.inc")).read()
    inc = rep(inc, "        return;\n    } else {", "        " + dump + "\n        return;\n    } else {")
    s = rep(s, '#include "tdnn_mx_epilogue.inc"\n}', inc + "\n" + dump + "\n}")
    s += """
extern "C" void ktf_prof_dump(void) {
    unsigned long long h[48][2][8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof2), sizeof(h));
    for (int i = 0; i < 48; ++i)
        for (int w = 0; w < 2; ++w)
            if (h[i][w][7]) {
                const double n = (double)h[i][w][7];
                printf("out %d nss %2d wave %d: %llu tiles | K-loop %.0f clk = %.2f us (%.0f MHz) | rest of the tile %.0f clk = %.2f us\\n",
                       i / 16, i % 16, w * 4, h[i][w][7], h[i][w][3] / n, h[i][w][4] / n / 100.0, 100.0 * (double)h[i][w][3] / (double)h[i][w][4],
                       h[i][w][6] / n, h[i][w][5] / n / 100.0);
            }
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof2), h, sizeof(h));
}
"""
    return s
V["prof2"] = _prof2

V["head"] = lambda s: subprocess.check_output(["git", "show", "HEAD:kaldi-tflite_amd/csrc/tdnn_mx.hip"], cwd=ROOT).decode()

# ring-depth experiment: the half-precision K-steps alone (no side DMAs / weight loads, no scaled MFMAs; WRONG results), on the
# committed two-slot kernel and on the working three-slot one
def _head():
    return subprocess.check_output(["git", "show", "HEAD:kaldi-tflite_amd/csrc/tdnn_mx.hip"], cwd=ROOT).decode()
def _f_only_common(s):
    s = rep(s, "                if (j == 0) {\n                    if (i == 3) MX_SA_SETUP(ss)", "                if (false) {\n                    if (i == 3) MX_SA_SETUP(ss)")
    return rep(s, "            for (int i = 0; i < 8; ++i) {\n                const u32x4 l = l_n, h = h_n;", "            for (int i = 0; i < 0; ++i) {\n                const u32x4 l = l_n, h = h_n;")
V["f_only_2slot"] = lambda s: rep(_f_only_common(_head()), "                if (j == 1) {\n                    if (i == 4) { MX_DMA_SW", "                if (false) {\n                    if (i == 4) { MX_DMA_SW")
# (the three-slot ring variants were built from a full copy of the kernel: kept as tools/mx/experiments/tdnn_mx_3slot.patch)
V["dma_only"] = V["no_mfma"]
V["dma_only_stage"] = lambda s: V["no_side_dma"](V["no_mfma"](s))
V["dma_only_sides"] = lambda s: V["no_f16_dma"](V["no_mfma"](s))
V["dma_only_none"] = lambda s: V["no_dma"](V["no_mfma"](s))



# A/B (correct results): the half-stage DMAs of K-step ks + 1 are all issued by ONE wave of each SIMD pair (waves w, w ^ 4) -- the pair
# takes turns K-step by K-step -- so that while one wave sits in vector-memory issue its partner has the matrix pipe to itself
def _alt_stage(s, sides=False):
    s = rep(s, "    MX_DMA_F16(0, 2) MX_DMA_F16(0, 3) MX_DMA_F16(0, 0) MX_DMA_F16(0, 1)",
            """#define MX_DMA_F16P(ks_, n_)                                                                                           \
    {                                                                                                                  \
        unsigned char* st_ = rsm + ((ks_) & 1) * MX_STAGE + (wave ^ 4) * 1024;                                         \
        if ((n_) < 2) {                                                                                                \
            int r_ = a_row[(n_) & 1] + ((wave & 4) ? -64 : 64) + f_off;                                                \
            r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                               \
            const unsigned vo_ = (f_base + (unsigned)r_) * 64u + a_cb[(n_) & 1];                                       \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xh + vo_), (lds_ptr_t*)(st_ + ((n_) & 1) * 8192), 16, 0, 0); \
        } else {                                                                                                       \
            const unsigned vo_ = (unsigned)(ks_) * (unsigned)MX_TILE + (unsigned)(((n_) & 1) * 512 + (tid ^ 256)) * 16u; \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wh + vo_), (lds_ptr_t*)(st_ + MX_TILE + ((n_) & 1) * 8192), 16, 0, 0); \
        }                                                                                                              \
    }
    MX_DMA_F16(0, 2) MX_DMA_F16(0, 3) MX_DMA_F16(0, 0) MX_DMA_F16(0, 1)""")
    s = rep(s, """                if (next) {
                    if (i == 0) MX_DMA_F16(ks + 1, 0)
                    if (i == 1) { MX_DMA_F16(ks + 1, 1) MX_F_ADV(ks + 2) }
                    if (i == 2) MX_DMA_F16(ks + 1, 2)
                    if (i == 3) MX_DMA_F16(ks + 1, 3)
                }""", """                if (next) {
                    const bool mine = ((wave >> 2) & 1) == (j & 1);
                    if (mine) {
                        if (i == 0) { MX_DMA_F16(ks + 1, 0) MX_DMA_F16P(ks + 1, 0) }
                        if (i == 1) { MX_DMA_F16(ks + 1, 1) MX_DMA_F16P(ks + 1, 1) }
                        if (i == 2) { MX_DMA_F16(ks + 1, 2) MX_DMA_F16P(ks + 1, 2) }
                        if (i == 3) { MX_DMA_F16(ks + 1, 3) MX_DMA_F16P(ks + 1, 3) }
                    }
                    if (i == 1) MX_F_ADV(ks + 2)
                }""")
    # the counted waits assumed every wave has the same DMAs in flight: wait for everything but this super-step's sides as before
    # (a wave that issued no stage has only side DMAs outstanding; vmcnt(6) / vmcnt(0) still cover what it must wait for)
    return s


V["alt_stage"] = _alt_stage
# ... spread over the whole K-step of the issuing wave (eight DMAs, one per row block) instead of two per row block in its first half
V["alt_stage_spread"] = lambda s: rep(_alt_stage(s), """                        if (i == 0) { MX_DMA_F16(ks + 1, 0) MX_DMA_F16P(ks + 1, 0) }
                        if (i == 1) { MX_DMA_F16(ks + 1, 1) MX_DMA_F16P(ks + 1, 1) }
                        if (i == 2) { MX_DMA_F16(ks + 1, 2) MX_DMA_F16P(ks + 1, 2) }
                        if (i == 3) { MX_DMA_F16(ks + 1, 3) MX_DMA_F16P(ks + 1, 3) }""", """                        if (i == 0) MX_DMA_F16(ks + 1, 0)
                        if (i == 1) MX_DMA_F16P(ks + 1, 0)
                        if (i == 2) MX_DMA_F16(ks + 1, 1)
                        if (i == 3) MX_DMA_F16P(ks + 1, 1)
                        if (i == 4) MX_DMA_F16(ks + 1, 2)
                        if (i == 5) MX_DMA_F16P(ks + 1, 2)
                        if (i == 6) MX_DMA_F16(ks + 1, 3)
                        if (i == 7) MX_DMA_F16P(ks + 1, 3)""")

# timing only (transposed results): every MFMA with its two operands exchanged -- what the register plane epilogue needs (a lane then
# holds four consecutive UNITS of one row). Is the exchange itself what cost the ported kernel 3 %?
V["swap_ops"] = lambda s: rep(rep(rep(s,
    "acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);",
    "acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[jj], a_cur, acc[i][jj], 0, 0, 0);"),
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al, bw, acc[i][jj], 4, 4, 0, asc, 0, wsc[jj]);",
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, al, acc[i][jj], 4, 4, 0, wsc[jj], 0, asc);"),
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah, bw, acc[i][jj], 4, 2, 1, asc, 1, wsc[jj]);",
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, ah, acc[i][jj], 2, 4, 1, wsc[jj], 1, asc);")
# ... only the half-precision MFMAs / only the block-scaled ones
V["swap_ops_f16"] = lambda s: rep(s,
    "acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);",
    "acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[jj], a_cur, acc[i][jj], 0, 0, 0);")
V["swap_ops_mx"] = lambda s: rep(rep(s,
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al, bw, acc[i][jj], 4, 4, 0, asc, 0, wsc[jj]);",
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, al, acc[i][jj], 4, 4, 0, wsc[jj], 0, asc);"),
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah, bw, acc[i][jj], 4, 2, 1, asc, 1, wsc[jj]);",
    "acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, ah, acc[i][jj], 2, 4, 1, wsc[jj], 1, asc);")

# timing only (WRONG results: the operand registers of the 16 x 16 tiles are fed to 32 x 32 instructions as they are): every MFMA in its
# 32 x 32 form -- v_mfma_f32_32x32x16_f16 and v_mfma_scale_f32_32x32x64_f8f6f4: the same operand bytes per lane, the same fragment reads,
# the same 128 accumulator registers, half as many matrix instructions of twice the length. Does the matrix SHAPE change what the DMA
# stream costs beside it (a 16 x 16 instruction holds the SIMD's vector issue for 8 of its 16 cycles, a 32 x 32 one for 8 of its 32)?
def _mfma32(s, f16=True, mx=True):
    s = rep(s, "    f32x4 acc[8][4];\n", "    typedef float f32x16_abl __attribute__((ext_vector_type(16)));\n    f32x16_abl accw[4][2];\n    f32x4 acc[8][4];\n")
    s = rep(s, "        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};",
            "        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};\n"
            "    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) accw[i][j][e] = 0.0f;")
    if f16:
        s = rep(s, "                    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);",
                "                    for (int jj = 0; jj < 2; ++jj) accw[i >> 1][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur, bh[(i & 1) * 2 + jj], accw[i >> 1][jj], 0, 0, 0);")
    if mx:
        s = rep(s, """                for (int jj = 0; jj < 4; ++jj) {     // residual of x (fp4, scale byte 0) times the fp4 image of w (scale byte 0)
                    const i32x8 bw = i32x8{(int)w4[jj].x, (int)w4[jj].y, (int)w4[jj].z, (int)w4[jj].w, 0, 0, 0, 0};
                    acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al, bw, acc[i][jj], 4, 4, 0, asc, 0, wsc[jj]);
                }""", """                for (int jj = 0; jj < 2; ++jj) {
                    const int jw = (i & 1) * 2 + jj;
                    const i32x8 bw = i32x8{(int)w4[jw].x, (int)w4[jw].y, (int)w4[jw].z, (int)w4[jw].w, 0, 0, 0, 0};
                    accw[i >> 1][jj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al, bw, accw[i >> 1][jj], 4, 4, 0, asc, 0, wsc[jw]);
                }""")
        s = rep(s, """                for (int jj = 0; jj < 4; ++jj) {     // fp4 image of x (scale byte 1) times the fp6 (e2m3) residual of w (scale byte 1)
                    const i32x8 bw = i32x8{(int)wl6a[jj].x, (int)wl6a[jj].y, (int)wl6a[jj].z, (int)wl6a[jj].w, (int)wl6b[jj].x, (int)wl6b[jj].y, 0, 0};
                    acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah, bw, acc[i][jj], 4, 2, 1, asc, 1, wsc[jj]);
                }""", """                for (int jj = 0; jj < 2; ++jj) {
                    const int jw = (i & 1) * 2 + jj;
                    const i32x8 bw = i32x8{(int)wl6a[jw].x, (int)wl6a[jw].y, (int)wl6a[jw].z, (int)wl6a[jw].w, (int)wl6b[jw].x, (int)wl6b[jw].y, 0, 0};
                    accw[i >> 1][jj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ah, bw, accw[i >> 1][jj], 4, 2, 1, asc, 1, wsc[jw]);
                }""")
    # the epilogue reads acc[8][4]: the 32 x 32 accumulators are added to it in place (sub-registers of the wide accumulators)
    s = rep(s, '#include "tdnn_mx_epilogue.inc"', """#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][jj][e] += accw[i >> 1][jj >> 1][((i & 1) * 2 + (jj & 1)) * 4 + e];
#include "tdnn_mx_epilogue.inc"
""")
    return s


V["mfma32"] = _mfma32
V["mfma32_no_dma"] = lambda s: V["no_dma"](_mfma32(s))      # (one shape at a time keeps both accumulator sets live: 190 spills)

# A/B (correct results): waves 4-7 (the SIMD partners of waves 0-3) issue their DMAs half a K-step away from their partners' -- the half
# stage between row blocks 4-7 instead of 0-3, the side pieces between row blocks 0-3 instead of 4-7 -- so that the two waves of a SIMD
# are never both in vector-memory issue (the K-loop exists twice, one copy per half of the workgroup)
def _stagger_dma(s):
    a = s.index("    for (int ss = 0; ss < p.nss; ++ss) {")
    b = s.index("#undef MX_DMA_F16")
    loop = s[a:b]
    sh = loop
    for k in range(8):
        sh = sh.replace(f"if (i == {k}) MX_DMA_F16", f"if (i == @{(k + 4) & 7}) MX_DMA_F16").replace(f"if (i == {k}) {{ MX_DMA_F16", f"if (i == @{(k + 4) & 7}) {{ MX_DMA_F16")
        sh = sh.replace(f"if (i == {k}) MX_DMA_S", f"if (i == @{(k + 4) & 7}) MX_DMA_S").replace(f"if (i == {k}) {{ MX_DMA_S", f"if (i == @{(k + 4) & 7}) {{ MX_DMA_S")
    sh = sh.replace("@", "").replace("if (i == 3) MX_SA_SETUP(ss)", "if (i == 0) MX_SA_SETUP(ss)")
    # (these waves issue a K-step's side pieces BEFORE its half stage: the stage is the youngest DMA at every barrier)
    sh = rep(sh, 'if (j == 1 || j == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");\n            else asm volatile', 'asm volatile')
    assert sh != loop
    return s[:a] + "    if (wave < 4) {\n" + loop + "    } else {\n" + sh + "    }\n" + s[b:]


V["stagger_dma"] = _stagger_dma

# A/B (correct results): cache policy of the LDS-DMA loads -- nt (aux = 2) on the activation pieces (half stage A + side A), on the weight pieces
# (half stage W + side W), or on both
def _nt(s, a, w):
    if a:
        s = rep(s, "(lds_ptr_t*)(st_ + ((n_) & 1) * 8192), 16, 0, 0);", "(lds_ptr_t*)(st_ + ((n_) & 1) * 8192), 16, 0, 2);")
        s = rep(s, "(lds_ptr_t*)(rsm + MX_SA_OFF + plane_ * 16384 + (kb_ * 256 + rg_ * 64) * 16), 16, 0, 0);", "(lds_ptr_t*)(rsm + MX_SA_OFF + plane_ * 16384 + (kb_ * 256 + rg_ * 64) * 16), 16, 0, 2);")
        s = rep(s, "(lds_ptr_t*)(rsm + MX_SA_OFF + 32768 + (kb_ * 256 + rg_ * 64) * 4), 4, 0, 0);", "(lds_ptr_t*)(rsm + MX_SA_OFF + 32768 + (kb_ * 256 + rg_ * 64) * 4), 4, 0, 2);")
    if w:
        s = rep(s, "(lds_ptr_t*)(st_ + MX_TILE + ((n_) & 1) * 8192), 16, 0, 0);", "(lds_ptr_t*)(st_ + MX_TILE + ((n_) & 1) * 8192), 16, 0, 2);")
        s = rep(s, "(lds_ptr_t*)(rsm + MX_SW_OFF + idx_ * 1024), 16, 0, 0);", "(lds_ptr_t*)(rsm + MX_SW_OFF + idx_ * 1024), 16, 0, 2);")
    return s


def _aux(s, v):
    s = _nt(s, True, True)
    return s.replace(", 16, 0, 2);", f", 16, 0, {v});").replace(", 4, 0, 2);", f", 4, 0, {v});")


V["aux_sc0"] = lambda s: _aux(s, 1)          # every LDS-DMA load with sc0 (aux bit 0) / sc1 (bit 4) / both
V["aux_sc1"] = lambda s: _aux(s, 16)
V["aux_sc0sc1"] = lambda s: _aux(s, 17)
V["nt_a"] = lambda s: _nt(s, True, False)
V["nt_w"] = lambda s: _nt(s, False, True)
V["nt_aw"] = lambda s: _nt(s, True, True)

names = sys.argv[1:] or list(V)
objs = [o for o in _product_objects() if o != "tdnn_mx.o"]
FLAGS = {
    "flags_O2": ["-O2"],
    "flags_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "flags_maxocc": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "flags_iter": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "flags_nounroll": ["-fno-unroll-loops"],
    "flags_noslp": ["-fno-slp-vectorize"],
    "flags_postra": ["-mllvm", "-enable-post-misched=false"],
}
for k in FLAGS:
    V[k] = lambda s: s
for n in names:
    path = f"/tmp/tdnn_mx_{n}.hip"
    open(path, "w").write(V[n](src))
    out = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", f"libktf_abl_{n}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I" + CS]
                          + FLAGS.get(n, []) + ["-c", path, "-o", f"/tmp/tdnn_mx_{n}.o"], cwd=CS)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(CS, o) for o in objs]
                          + [f"/tmp/tdnn_mx_{n}.o", "-o", out])
    print("built", out)
