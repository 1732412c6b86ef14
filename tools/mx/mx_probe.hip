// Probe (measurement tool, not part of the library): operand layout and scale semantics of
// v_mfma_scale_f32_16x16x128_f8f6f4 with fp4 (e2m1) A and fp4 / fp6 (e2m3) B operands on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 mx_probe.hip -o mx_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int BFMT, int OPA, int OPB>   // BFMT: 4 = fp4 B, 2 = e2m3 B
__global__ void k(const uint32_t* a, const uint32_t* b, const uint32_t* sa, const uint32_t* sb, float* c) {
    int lane = threadIdx.x;
    i32x8 av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) av[i] = a[lane * 8 + i];
    for (int i = 0; i < (BFMT == 4 ? 4 : 6); ++i) bv[i] = b[lane * 8 + i];
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 4, BFMT, OPA, sa[lane], OPB, sb[lane]);
    for (int r = 0; r < 4; ++r) c[lane * 4 + r] = acc[r];
}

static float fp4(int code) {
    static const float t[8] = {0, 0.5f, 1, 1.5f, 2, 3, 4, 6};
    return (code & 8) ? -t[code & 7] : t[code & 7];
}
static float e2m3(int code) {
    int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
    float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
    return s ? -v : v;
}

int main() {
    const int L = 64;
    uint32_t ha[L * 8], hb[L * 8], hsa[L], hsb[L];
    float hc[L * 4];
    uint32_t *da, *db, *dsa, *dsb;
    float* dc;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dc, sizeof hc);
    srand(7);
    for (int bfmt = 4; bfmt >= 2; bfmt -= 2)
        for (int op = 0; op < 4; op += 3) {
            for (int i = 0; i < L * 8; ++i) { ha[i] = ((uint32_t)rand() << 16) ^ rand(); hb[i] = ((uint32_t)rand() << 16) ^ rand(); }
            for (int i = 0; i < L; ++i) {
                hsa[i] = 0; hsb[i] = 0;
                for (int by = 0; by < 4; ++by) { hsa[i] |= (uint32_t)(120 + rand() % 12) << (8 * by); hsb[i] |= (uint32_t)(120 + rand() % 12) << (8 * by); }
            }
            hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
            hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
            if (bfmt == 4 && op == 0) k<4, 0, 0><<<1, 64>>>(da, db, dsa, dsb, dc);
            if (bfmt == 4 && op == 3) k<4, 3, 1><<<1, 64>>>(da, db, dsa, dsb, dc);
            if (bfmt == 2 && op == 0) k<2, 0, 0><<<1, 64>>>(da, db, dsa, dsb, dc);
            if (bfmt == 2 && op == 3) k<2, 2, 3><<<1, 64>>>(da, db, dsa, dsb, dc);
            hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
            const int opa = (op == 0) ? 0 : (bfmt == 4 ? 3 : 2), opb = (op == 0) ? 0 : (bfmt == 4 ? 1 : 3);
            // hypothesis: lane l = (row/col l&15, K-block l>>4); element e of the block at bit offset e*bits; scale = 2^(byte[opsel] - 127) per lane;
            // acc[r] of lane l = C[(l>>4)*4 + r][l&15]
            double maxd = 0, maxv = 0;
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double s = 0;
                    for (int q = 0; q < 4; ++q) {
                        const int la = q * 16 + m, lb = q * 16 + n;
                        const double sca = ldexp(1.0, (int)((hsa[la] >> (8 * opa)) & 255) - 127), scb = ldexp(1.0, (int)((hsb[lb] >> (8 * opb)) & 255) - 127);
                        double blk = 0;
                        for (int e = 0; e < 32; ++e) {
                            const int ca = (ha[la * 8 + e / 8] >> (4 * (e % 8))) & 15;
                            double vb;
                            if (bfmt == 4) vb = fp4((hb[lb * 8 + e / 8] >> (4 * (e % 8))) & 15);
                            else {
                                const int bit = 6 * e;
                                uint64_t w = (uint64_t)hb[lb * 8 + bit / 32] | ((uint64_t)hb[lb * 8 + bit / 32 + 1] << 32);
                                vb = e2m3((int)((w >> (bit % 32)) & 63));
                            }
                            blk += (double)fp4(ca) * vb;
                        }
                        s += blk * sca * scb;
                    }
                    const int lane = (m >> 2) * 16 + n, r = m & 3;
                    const double d = fabs(s - hc[lane * 4 + r]);
                    if (d > maxd) maxd = d;
                    if (fabs(s) > maxv) maxv = fabs(s);
                }
            printf("B format %s, opsel a/b %d/%d: max |gpu - cpu| = %.3e (max |value| %.3e)\n", bfmt == 4 ? "fp4" : "e2m3", opa, opb, maxd, maxv);
        }
    return 0;
}
