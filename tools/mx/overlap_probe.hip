// Do LDS-DMA instructions and MFMAs overlap on a CU? (measurement tool, not product)
// 8 "matrix" waves (2 per SIMD) each run `iters` rounds of 32 independent v_mfma_f32_16x16x32_f16 on register operands. The same
// number of 1 KiB global_load_lds_dwordx4 instructions per round is then issued (a) by nobody, (b) by the matrix waves themselves,
// one after every few MFMAs, (c) by 4 extra loader waves that do nothing else. Reported: ns per round and CU.
//   hipcc -O3 --offload-arch=gfx950 tools/mx/overlap_probe.hip -o tools/mx/overlap_probe && tools/mx/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// mode 0: no DMA; 1: matrix waves issue `dma_per_wave` DMAs per round; 2: loader waves issue 8 * dma_per_wave / 4 each per round
template <int NM>
__global__ __launch_bounds__(NM == 16 ? 1024 : 768) void probe(const char* src, int mode, int dma_per_wave, int iters, long long* cyc, float* sink, int frag) {
    constexpr int MPW = 256 / NM;                        // MFMAs per matrix wave and round: the CU's MFMA work per round is fixed
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (size_t)(blockIdx.x & 7) * 65536;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (wave < NM) {
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        const int every = (mode == 1 && dma_per_wave > 0) ? MPW / dma_per_wave : 1 << 30;
        const h8* lfrag = reinterpret_cast<const h8*>(lds) + lane;          // conflict-free 16-byte fragment reads
        for (int it = 0; it < iters; ++it) {
            h8 bf[4] = {b, b, b, b};
            if (frag) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bf[jj] = lfrag[((it + jj) & 7) * 64 + 2048];
            }
#pragma unroll
            for (int m = 0; m < MPW; ++m) {
                if (frag && (m & 3) == 0) a = lfrag[((it + wave + (m >> 2)) & 31) * 64];     // the row block's A fragment (as in the GEMM: 1 read per 4 MFMAs)
                acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[m & 3], acc[m & 7], 0, 0, 0);
                if (mode == 1 && (m % every) == every - 1) {
                    const unsigned off = (unsigned)(((it * MPW + m) * NM + wave) & 63) * 1024u + lane * 16u;
                    __builtin_amdgcn_global_load_lds((glb_ptr_t*)(base + off), (lds_ptr_t*)(lds + ((m * NM + wave) & 63) * 1024), 16, 0, 0);
                }
            }
            if (mode == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.f) sink[0] = s;
    } else if (mode == 2) {
        const int per_round = dma_per_wave * NM / 4;     // 4 loader waves carry what the matrix waves would
        for (int it = 0; it < iters; ++it) {
            for (int k = 0; k < per_round; ++k) {
                const unsigned off = (unsigned)(((it * per_round + k) * 4 + (wave - NM)) & 63) * 1024u + lane * 16u;
                __builtin_amdgcn_global_load_lds((glb_ptr_t*)(base + off), (lds_ptr_t*)(lds + ((k * 4 + wave) & 63) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = wall_clock64() - t0;
}

int main() {
    char* src; long long* cyc; float* sink;
    hipMalloc(&src, 8 * 65536); hipMemset(src, 0, 8 * 65536);
    hipMalloc(&cyc, 256 * sizeof(long long)); hipMalloc(&sink, 16);
    hipFuncSetAttribute((const void*)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<long long> h(256);
    const int iters = 2000;
    for (int frag : {0, 1})
    for (int nm : {8, 16})
        for (int per_cu : {0, 16, 32, 64, 128})
            for (int mode = 0; mode < 3; ++mode) {
                if ((per_cu == 0) != (mode == 0)) continue;
                if (nm == 16 && mode == 2) continue;             // (16 matrix waves fill the workgroup: no room for loader waves)
                const int dpw = per_cu / nm;
                if (mode != 0 && dpw == 0) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    if (nm == 8) probe<8><<<256, 768, 65536>>>(src, mode, dpw, iters, cyc, sink, frag);
                    else probe<16><<<256, 1024, 65536>>>(src, mode, dpw, iters, cyc, sink, frag);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
                double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
                printf("%s%2d matrix waves x %2d MFMAs, %3d DMA per round and CU, %s: %.1f ns per round (256 MFMAs per CU = 1024 clk per SIMD)\n",
                       frag ? "operands from LDS (12 / 8 fragment reads per wave and round): " : "register operands: ", nm, 256 / nm, per_cu, mode == 0 ? "no DMA                    " : mode == 1 ? "issued by the matrix waves" : "issued by 4 loader waves  ", mean * 10.0 / iters);
            }
    return 0;
}
