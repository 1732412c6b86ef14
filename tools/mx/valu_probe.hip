// Probe (measurement tool, not part of the library): issue cost of the vector instructions the f16mx plane encoder is made of, on
// gfx950: shader clocks per wave-instruction for a stream of N independent instructions, one wave and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 valu_probe.hip -o valu_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void k(const float* in, float* out, long long* clk) {
    const int lane = threadIdx.x & 63;
    float a = in[lane], b = in[lane + 64], s = in[128];
    if (KIND >= 5) { s = __uint_as_float((unsigned)(100 + (lane * 7) % 40) << 23); a *= 1e-4f * (1 + lane % 5); b *= 3e-5f * (1 + lane % 3); }
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    h2 hh = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){a, b}, h2);
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; ++it) {
        if (KIND == 0 || KIND == 5) {          // 8 independent dwords, bytes 0..3 of each in turn (4 dependent steps per dword)
            REP8(r0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r0, a, b, s, 0); r1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r1, a, b, s, 1);
                 r2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r2, a, b, s, 2); r3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r3, a, b, s, 3);
                 r4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r4, b, a, s, 0); r5 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r5, b, a, s, 1);
                 r6 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r6, b, a, s, 2); r7 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r7, b, a, s, 3);)
        } else if (KIND == 1 || KIND == 6) {
            REP8(r0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r0, hh, s, 0); r1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r1, hh, s, 1);
                 r2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r2, hh, s, 2); r3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r3, hh, s, 3);
                 r4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r4, hh, s, 0); r5 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r5, hh, s, 1);
                 r6 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r6, hh, s, 2); r7 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r7, hh, s, 3);)
        } else if (KIND == 2) {   // permlane swaps
            REP8({ auto q = __builtin_amdgcn_permlane16_swap(ua, ub, false, false); ua = q[0]; ub = q[1]; }
                 { auto q = __builtin_amdgcn_permlane32_swap(r0, r1, false, false); r0 = q[0]; r1 = q[1]; }
                 { auto q = __builtin_amdgcn_permlane16_swap(r2, r3, false, false); r2 = q[0]; r3 = q[1]; }
                 { auto q = __builtin_amdgcn_permlane32_swap(r4, r5, false, false); r4 = q[0]; r5 = q[1]; }
                 { auto q = __builtin_amdgcn_permlane16_swap(r6, r7, false, false); r6 = q[0]; r7 = q[1]; }
                 { auto q = __builtin_amdgcn_permlane32_swap(ua, ub, false, false); ua = q[0]; ub = q[1]; }
                 { auto q = __builtin_amdgcn_permlane16_swap(r0, r1, false, false); r0 = q[0]; r1 = q[1]; }
                 { auto q = __builtin_amdgcn_permlane32_swap(r2, r3, false, false); r2 = q[0]; r3 = q[1]; })
        } else if (KIND == 3) {   // plain fma, 8 independent chains
            float f0 = a, f1 = b, f2 = a, f3 = b, f4 = a, f5 = b, f6 = a, f7 = b;
            REP8(f0 = fmaf(f0, s, a); f1 = fmaf(f1, s, b); f2 = fmaf(f2, s, a); f3 = fmaf(f3, s, b); f4 = fmaf(f4, s, a); f5 = fmaf(f5, s, b); f6 = fmaf(f6, s, a); f7 = fmaf(f7, s, b);)
            r0 ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
        } else if (KIND == 4) {   // v_cvt_pk_f16_f32
            REP8(r0 ^= __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){a + __uint_as_float(r1), b}, h2));
                 r1 ^= __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){b + __uint_as_float(r2), a}, h2));
                 r2 ^= __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){a + __uint_as_float(r3), b}, h2));
                 r3 ^= __builtin_bit_cast(unsigned, __builtin_convertvector((__attribute__((ext_vector_type(2))) float){b + __uint_as_float(r0), a}, h2));)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ ua ^ ub);
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
    float *in, *out;
    long long* clk;
    hipMalloc(&in, 4096); hipMalloc(&out, 1 << 20); hipMalloc(&clk, 64);
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0.37f * (i % 17) - 2.0f;
    h[128] = 0.5f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const char* names[7] = {"v_cvt_scalef32_pk_fp4_f32 (1024 per wave)", "v_cvt_scalef32_pk_fp4_f16 (1024 per wave)", "v_permlane16/32_swap (1024 per wave)",
                            "v_fma_f32 (1024 per wave + 8)", "v_cvt_pk_f16_f32 + add + xor (512 triples per wave)",
                            "v_cvt_scalef32_pk_fp4_f32, per-lane scales, small values", "v_cvt_scalef32_pk_fp4_f16, per-lane scales, small values"};
    for (int kind = 0; kind < 7; ++kind)
        for (int waves = 4; waves <= 16; waves *= 2) {      // one block per CU slot: 4 / 8 / 16 waves = 1 / 2 / 4 per SIMD
            long long c = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) k<0><<<1, waves * 64>>>(in, out, clk);
                if (kind == 1) k<1><<<1, waves * 64>>>(in, out, clk);
                if (kind == 2) k<2><<<1, waves * 64>>>(in, out, clk);
                if (kind == 3) k<3><<<1, waves * 64>>>(in, out, clk);
                if (kind == 4) k<4><<<1, waves * 64>>>(in, out, clk);
                if (kind == 5) k<5><<<1, waves * 64>>>(in, out, clk);
                if (kind == 6) k<6><<<1, waves * 64>>>(in, out, clk);
                hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            }
            printf("%-52s %2d waves/CU: %8lld clk = %.1f clk per wave-instruction of wave 0\n", names[kind], waves, c, c / 1024.0);
        }
    return 0;
}
