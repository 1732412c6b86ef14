#!/usr/bin/env python3
"""Phase stamps for the 256-row f16mx kernel (measurement tool, not the product): patches a COPY of csrc/tdnn_mx.hip -- this round's or an
earlier one's (git show <rev>:kaldi-tflite_amd/csrc/tdnn_mx.hip) -- so that thread 0 of every workgroup adds the 100 MHz wall clock to four
accumulators per (output kind, super-steps): kernel entry, first K-loop barrier passed, K-loop done, tile done. Differences of the sums are the
summed phase times (the bench workload launches no empty tile). `ktf_prof_dump()` prints and clears them.

    python tools/mx/prof_stamps.py <in.hip> <out.hip>;  tools/build_variant.sh prof <out.hip name in csrc>   (see tools/mx/prof_run.py)
"""
import re, sys
s = open(sys.argv[1]).read()
head = '''#include "tdnn_mx_common.h"
__device__ unsigned long long g_prof[48][8];
__device__ __forceinline__ void mx_stamp(int slot, int which) {
    if (threadIdx.x == 0) {
        atomicAdd(&g_prof[slot][which], (unsigned long long)wall_clock64());
        if (which == 0) atomicAdd(&g_prof[slot][4], 1ull);
    }
}
extern "C" void ktf_prof_dump(void) {
    unsigned long long h[48][8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h));
    for (int i = 0; i < 48; ++i)
        if (h[i][4]) {
            const double n = (double)h[i][4] * 100.0;
            printf("out %d nss %2d: %llu tiles | entry -> first stage issued %.2f us -> thread 0 at the first wait %.2f us -> barrier passed %.2f us | prologue %.2f us | K-loop %.2f us (%.3f per super-step) | epilogue %.2f us (thread 0 done %.2f us earlier) | tile %.2f us\\n", i / 16, i % 16, h[i][4],
                   (double)(long long)(h[i][5] - h[i][0]) / n, (double)(long long)(h[i][6] - h[i][5]) / n, (double)(long long)(h[i][1] - h[i][6]) / n,
                   (double)(long long)(h[i][1] - h[i][0]) / n, (double)(long long)(h[i][2] - h[i][1]) / n, (double)(long long)(h[i][2] - h[i][1]) / n / (i % 16),
                   (double)(long long)(h[i][3] - h[i][2]) / n, (double)(long long)(h[i][3] - h[i][7]) / n, (double)(long long)(h[i][3] - h[i][0]) / n);
        }
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), h, sizeof(h));
}
#define MX_PROF_SLOT ((p.nss < 15 ? p.nss : 15) + 16 * OUT)
'''
assert s.count('#include "tdnn_mx_common.h"') == 1
s = s.replace('#include "tdnn_mx_common.h"', head)
# T1: behind the K-loop's barrier, first K-step only
s, n = re.subn(r'(__builtin_amdgcn_s_barrier\(\);\s*\n\s*asm volatile\("" ::: "memory"\);)', r'\1\n                if (ks == 0) mx_stamp(MX_PROF_SLOT, 1);', s)
assert n == 1, n
# T5 / T6: the first stage's DMAs issued; thread 0 arrives at the K-loop's first wait
if "MX_DMA_A_E(0, 0, 1)\n" in s:
    s = s.replace("MX_DMA_A_E(0, 0, 1)\n", "MX_DMA_A_E(0, 0, 1)\n    mx_stamp(MX_PROF_SLOT, 5);\n", 1)
else:
    assert s.count("    MX_F_ADV(1)\n") == 1
    s = s.replace("    MX_F_ADV(1)\n", "    MX_F_ADV(1)\n    mx_stamp(MX_PROF_SLOT, 5);\n")
s, n = re.subn(r'(\n\s*)(if \(j == 1 \|\| j == 2\) asm volatile\("s_waitcnt vmcnt\(6\)")', r'\1if (ks == 0) mx_stamp(MX_PROF_SLOT, 6);\1\2', s)
assert n == 1, n
# T2: in front of the epilogue
assert s.count('#include "tdnn_mx_epilogue.inc"') == 1
s = s.replace('#include "tdnn_mx_epilogue.inc"', 'mx_stamp(MX_PROF_SLOT, 2);\n#include "tdnn_mx_epilogue.inc"')
# T0 / T3: around the tile body in the kernel
s, n = re.subn(r'(\n    mx_tile<ACT, OUT, PADK, FLAT>\(p, blockIdx\.x, mtiles, ntiles, gtiles, stats, rsm\);)',
               r'\n    mx_stamp(MX_PROF_SLOT, 0);\1\n    __builtin_amdgcn_s_waitcnt(0);\n    mx_stamp(MX_PROF_SLOT, 7);\n    __syncthreads();\n    mx_stamp(MX_PROF_SLOT, 3);', s)
assert n == 1, n
open(sys.argv[2], "w").write(s)
