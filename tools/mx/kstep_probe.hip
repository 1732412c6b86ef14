// What does the LDS read latency behind a K-step's barrier cost the f16mx tile, and would the scaled MFMAs of the previous super-step,
// issued right behind the barrier, hide it? (measurement tool, not product: the instruction mix of one half-precision K-step, no results)
// 8 waves (two per SIMD, 128 x 64 each), per round: barrier -> 4 W fragments + the first A fragment (ds_read_b128; their data "landed" with
// the barrier, so they cannot be read earlier) -> 32 v_mfma_f32_16x16x32_f16 with the next A fragment read under each group of four -> 7
// LDS-DMA instructions per wave between the groups (L2-resident 2 MiB window) -> s_waitcnt vmcnt(0).
//   FILL = 0: as the kernel today; the 16 scaled MFMAs per wave and K-step (64 per super-step) run in a phase of their own: + 16 x 16 clk
//             per wave, i.e. + 2 x 256 clk per SIMD and K-step on top of the measured round (reported as "+ M").
//   FILL = 16: 16 register-operand MFMAs (standing in for v_mfma_scale_f32_16x16x128_f8f6f4, also 16 clk) right behind the fragment reads.
//   hipcc -O3 --offload-arch=gfx950 tools/mx/kstep_probe.hip -o tools/mx/kstep_probe && tools/mx/kstep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int FILL, int DPW>
__global__ __launch_bounds__(512) void probe(const char* src, int iters, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* gsrc = src + (size_t)wave * 1024 + lane * 16;
    unsigned char* ldst = lds + 64 * 1024 + wave * 1024;
    f4 acc[32], accm[8];
    for (int i = 0; i < 32; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) accm[i] = f4{0.f, 0.f, 0.f, 0.f};
    h8 ra, rb;
    for (int i = 0; i < 8; ++i) { ra[i] = (_Float16)(0.001f * (lane + i)); rb[i] = (_Float16)(0.002f * (lane - i)); }
    const h8* lfrag = reinterpret_cast<const h8*>(lds) + lane;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (DPW > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        h8 bf[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bf[jj] = lfrag[((it + jj) & 7) * 64 + 2048];
        h8 a_cur = lfrag[((it + wave) & 31) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < FILL; ++f) accm[f & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra, rb, accm[f & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h8 a_nxt = a_cur;
            if (i < 7) a_nxt = lfrag[((it + wave + i + 1) & 31) * 64];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bf[jj], acc[i * 4 + jj], 0, 0, 0);
            a_cur = a_nxt;
            __builtin_amdgcn_sched_barrier(0);
            if (i < DPW)
                __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gsrc + (size_t)(((it * 8 + i) & 31) * 65536 + (blockIdx.x & 7) * 8192)), (lds_ptr_t*)(ldst + (i & 3) * 8192), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += accm[i][0];
    if (s == 12345.f) sink[0] = s;
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = wall_clock64() - t0;
}

template <int FILL, int DPW>
static double run(const char* src, long long* cyc, float* sink) {
    const int iters = 4000;
    (void)hipFuncSetAttribute((const void*)probe<FILL, DPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    std::vector<long long> h(256);
    for (int rep = 0; rep < 3; ++rep) {
        probe<FILL, DPW><<<256, 512, 150 * 1024>>>(src, iters, cyc, sink);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
    return mean * 10.0 / iters;
}

int main() {
    char* src; long long* cyc; float* sink;
    (void)hipMalloc(&src, 4 << 20); (void)hipMemset(src, 0, 4 << 20);
    (void)hipMalloc(&cyc, 256 * sizeof(long long)); (void)hipMalloc(&sink, 16);
    for (int rep = 0; rep < 2; ++rep) {
        const double a0 = run<0, 0>(src, cyc, sink), a7 = run<0, 7>(src, cyc, sink);
        const double m16 = run<16, 0>(src, cyc, sink) - 0.0, b7 = run<16, 7>(src, cyc, sink);
        const double monly = run<16, 0>(src, cyc, sink) - a0;       // what 16 MFMAs per wave add when they fill the read latency
        printf("K-step, no DMA: %.1f ns; with 7 DMAs per wave: %.1f ns\n", a0, a7);
        printf("  + 16 MFMAs per wave behind the barrier's fragment reads: %.1f ns without DMA (+ %.1f), %.1f ns with (+ %.1f); the same 16 MFMAs as a phase of their own: + %.1f ns at 2.1 GHz\n",
               m16, monly, b7, b7 - a7, 2 * 16 * 16 / 2.1);
    }
    return 0;
}
