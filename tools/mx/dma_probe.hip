// LDS-DMA issue-rate probe (measurement tool, not product): clocks per 1 KiB `global_load_lds_dwordx4` / `buffer_load_dwordx4 ... lds`
// instruction on one CU, by how many of a workgroup's 8 waves issue them, from an L2-resident source. Every CU runs one workgroup.
//   hipcc -O3 --offload-arch=gfx950 tools/mx/dma_probe.hip -o tools/mx/dma_probe && tools/mx/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, int waves_issuing, int per_wave, int reps, long long* cyc, int src_kib) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (size_t)(blockIdx.x % 8) * src_kib * 1024;     // a few distinct L2-resident regions
    __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, src_kib * 1024, 0x00020000);   // raw buffer
    __syncthreads();
    const long long t0 = wall_clock64();
    if (wave < waves_issuing) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll 4
            for (int k = 0; k < per_wave; ++k) {
                const unsigned off = (unsigned)(((r * per_wave + k) * 8 + wave) % src_kib) * 1024u + lane * 16u;
                unsigned char* dst = lds + ((k * 8 + wave) % 128) * 1024;
                if (FORM == 0) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(base + off), (lds_ptr_t*)dst, 16, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t*)dst, 16, off, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = wall_clock64() - t0;
}

int main() {
    const int src_kib = 64;
    char* src; long long* cyc;
    hipMalloc(&src, (size_t)8 * src_kib * 1024);
    hipMemset(src, 1, (size_t)8 * src_kib * 1024);
    hipMalloc(&cyc, 256 * sizeof(long long));
    hipFuncSetAttribute((const void*)dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    std::vector<long long> h(256);
    for (int form = 0; form < 2; ++form)
        for (int grid : {1, 256})
            for (int wi : {1, 2, 4, 8}) {
                const int per_wave = 16, reps = 200;
                for (int it = 0; it < 2; ++it) {
                    if (form == 0) dma_kernel<0><<<grid, 512, 131072>>>(src, wi, per_wave, reps, cyc, src_kib);
                    else dma_kernel<1><<<grid, 512, 131072>>>(src, wi, per_wave, reps, cyc, src_kib);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
                double mean = 0;
                for (int i = 0; i < grid; ++i) mean += (double)h[i];
                mean /= grid;
                const double n = (double)wi * per_wave * reps;       // DMA instructions per CU
                printf("%s, %3d CUs, %d waves issuing: %.1f ns per 1 KiB DMA instruction per CU = %.1f GB/s per CU\n",
                       form ? "buffer_load lds" : "global_load_lds", grid, wi, mean * 10.0 / n, 1024.0 / (mean * 10.0 / n));
            }
    return 0;
}
