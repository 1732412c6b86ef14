// Would FOUR matrix waves (one per SIMD, 128 x 128 of a 256 x 256 tile each, 512 registers) run the K-step of the f16mx tile faster than
// the kernel's EIGHT (two per SIMD, 128 x 64 each)? (measurement tool, not product: no results, only the instruction mix of one K-step)
// Per round and CU: 256 v_mfma_f32_16x16x32_f16 (one 32-deep K-step of the half-precision pass), the fragment reads they need from LDS
// (8 waves: 8 A + 4 W ds_read_b128 per wave = 96; 4 waves: 8 A + 8 W = 64; read one round ahead into a second set of registers) and 0 / 56 LDS-DMA instructions of 1 KiB (the kernel's 224 per
// four K-steps) issued by the matrix waves between their MFMAs. Reported: ns per round.
//   hipcc -O3 --offload-arch=gfx950 tools/mx/wave4_probe.hip -o tools/mx/wave4_probe && tools/mx/wave4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NM, int DPW, int STREAM = 0, int TIGHT = 0>      // TIGHT 1: a round's DMAs (issued in its first half) have landed before its barrier -- the
                                                                // kernel's two-stage ring: one K-step of lookahead; 0: one round more. STREAM 1: the DMA source walks a 2 MiB window (L2 hits, not L1 hits); DPW: LDS-DMA instructions per wave and round (compile time: a run-time cadence costs a scalar branch per MFMA)
__global__ __launch_bounds__(NM * 64) void probe(const char* src, int iters, long long* cyc, float* sink) {
    constexpr int MPW = 256 / NM;                        // MFMAs per wave and round
    constexpr int NB = NM == 4 ? 8 : (NM == 8 ? 4 : 2);  // W fragments per wave (16-column blocks); A fragments: 8 (128 rows)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (STREAM ? 0 : (size_t)(blockIdx.x & 7) * 65536);
    __syncthreads();
    const long long t0 = wall_clock64();
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    constexpr int every = DPW > 0 ? MPW / DPW : 1 << 30;
    const char* gsrc = base + (size_t)wave * 1024 + lane * 16;       // a wave's DMAs: fixed 1 KiB pieces of an L2-resident buffer
    unsigned char* ldst = lds + 64 * 1024 + wave * 1024;
    const h8* lfrag = reinterpret_cast<const h8*>(lds) + lane;          // conflict-free 16-byte fragment reads
    // fragments double-buffered in registers: the reads of round it + 1 are spread between the MFMAs of round it (a wave alone on its SIMD
    // has no other wave to hide an LDS read behind; 2 x (8 + NB) x 4 registers: 128 for four waves, beside 256 accumulator registers)
    h8 af[2][8], bf[2][NB];
#pragma unroll
    for (int k = 0; k < 8; ++k) af[0][k] = lfrag[((wave + k) & 31) * 64];
#pragma unroll
    for (int k = 0; k < NB; ++k) bf[0][k] = lfrag[(k & 7) * 64 + 2048];
    constexpr int NF = 8 + NB, FSTEP = MPW / NF > 0 ? MPW / NF : 1;
#define ROUND(CUR, NXT)                                                                                                \
    {                                                                                                                  \
        _Pragma("unroll") for (int m = 0; m < MPW; ++m) {                                                              \
            if (m % FSTEP == 0 && m / FSTEP < NF) {                                                                    \
                const int k = m / FSTEP;                                                                               \
                if (k < 8) af[NXT][k] = lfrag[((it + 1 + wave + k) & 31) * 64];                                        \
                else bf[NXT][k - 8] = lfrag[((it + 1 + k) & 7) * 64 + 2048];                                           \
            }                                                                                                          \
            acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[CUR][m / NB], bf[CUR][m % NB], acc[m & 15], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);          /* keep the reads and DMAs where they are written: between the MFMAs */ \
            if constexpr (DPW > 0) {                                                                                   \
                if ((m % (TIGHT ? every / 2 : every)) == (TIGHT ? every / 2 : every) - 1 && m / (TIGHT ? every / 2 : every) < DPW)                                                                          \
                    __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gsrc + (STREAM ? (size_t)(((it * DPW + m / every) & 31) * 65536 + (blockIdx.x & 7) * 8192) : (size_t)0) + ((m / every) & 3) * NM * 1024), \
                                                     (lds_ptr_t*)(ldst + ((m / every) & 3) * NM * 1024), 16, 0, 0);     \
            }                                                                                                          \
        }                                                                                                              \
        if (DPW > 0) { if (TIGHT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }                                         \
        __builtin_amdgcn_s_barrier();                                                                                  \
    }
    for (int it = 0; it < iters; it += 2) {
        ROUND(0, 1)
        ++it;
        ROUND(1, 0)
        --it;
    }
#undef ROUND
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 12345.f) sink[0] = s;
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = wall_clock64() - t0;
}

template <int NM, int DPW, int STREAM = 0, int TIGHT = 0>
static void run(const char* src, long long* cyc, float* sink) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)probe<NM, DPW, STREAM, TIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);   // one workgroup per CU, as the kernel
    std::vector<long long> h(256);
    for (int rep = 0; rep < 3; ++rep) {
        probe<NM, DPW, STREAM, TIGHT><<<256, NM * 64, 150 * 1024>>>(src, iters, cyc, sink);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
    printf("%2d matrix waves x %2d MFMAs, %2d fragment reads per wave, %3d DMA per round and CU%s: %.1f ns per round\n", NM, 256 / NM,
           8 + (NM == 4 ? 8 : NM == 8 ? 4 : 2), DPW * NM, STREAM ? (TIGHT ? " from a 2 MiB window (L2), landed within the round" : " from a 2 MiB window (L2)") : DPW ? " from 16 / 32 KiB (L1)" : "", mean * 10.0 / iters);
}

int main() {
    char* src; long long* cyc; float* sink;
    hipMalloc(&src, 4 << 20); hipMemset(src, 0, 4 << 20);
    hipMalloc(&cyc, 256 * sizeof(long long)); hipMalloc(&sink, 16);
    run<4, 0>(src, cyc, sink); run<8, 0>(src, cyc, sink);
    run<4, 8>(src, cyc, sink); run<8, 4>(src, cyc, sink);                 // 32 per round and CU
    run<4, 16>(src, cyc, sink); run<8, 8>(src, cyc, sink);                // 64 (the kernel: 56)
    run<4, 8, 1>(src, cyc, sink); run<8, 4, 1>(src, cyc, sink);
    run<4, 16, 1>(src, cyc, sink); run<8, 8, 1>(src, cyc, sink);
    run<4, 16, 1, 1>(src, cyc, sink); run<8, 8, 1, 1>(src, cyc, sink);
    return 0;
}
