// Store-throughput probe (measurement tool, not product): what bounds the f16mx epilogue's 200 KiB of plane stores per 256x256 tile?
// One 512-thread workgroup per CU writes `kib` KiB as 1 KiB-per-instruction dwordx4 stores (the epilogue's shape), `reps` times into
// fresh addresses; variants: store policy (plain / nt), how many CUs store at the same time, bytes per burst.
//   hipcc -O3 --offload-arch=gfx950 tools/mx/store_probe.hip -o tools/mx/store_probe && tools/mx/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* out, int kib, int reps, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long wg_bytes = (long long)kib * 1024;
    const u32x4 v = {(unsigned)threadIdx.x, (unsigned)blockIdx.x, 3u, 4u};
    long long t_sum = 0;
    for (int r = 0; r < reps; ++r) {
        unsigned char* base = out + ((long long)r * gridDim.x + blockIdx.x) * wg_bytes;
        __syncthreads();
        const long long t0 = wall_clock64();
        for (int k = wave; k < kib; k += 8) {
            u32x4* p = reinterpret_cast<u32x4*>(base + (long long)k * 1024) + lane;
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
        __builtin_amdgcn_s_waitcnt(0);                   // vmcnt(0): this wave's stores are acknowledged
        __syncthreads();
        t_sum += wall_clock64() - t0;
        // a stretch of ALU work between bursts (a stand-in for the K-loop), so that bursts of different CUs are independent
        float a = (float)lane;
        for (int i = 0; i < 4000; ++i) a = a * 1.0001f + 0.5f;
        if (a == 12345.f) out[0] = 1;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t_sum;
}

int main() {
    const int kib = 200, reps = 16;
    unsigned char* out; long long* cyc;
    hipMalloc(&out, (size_t)256 * reps * kib * 1024 + 4096);
    hipMalloc(&cyc, 256 * sizeof(long long));
    std::vector<long long> h(256);
    for (int nt = 0; nt < 2; ++nt)
        for (int grid : {1, 8, 32, 64, 128, 256}) {
            for (int it = 0; it < 3; ++it) {
                if (nt) store_kernel<1><<<grid, 512>>>(out, kib, reps, cyc); else store_kernel<0><<<grid, 512>>>(out, kib, reps, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
            double mean = 0, mx = 0;
            for (int i = 0; i < grid; ++i) { mean += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
            mean /= grid * reps; mx /= reps;
            // wall_clock64 ticks at 100 MHz
            printf("%s grid %3d: burst of %d KiB per CU takes mean %.2f us (max %.2f us) = %.1f GB/s per CU, %.2f TB/s over the grid\n",
                   nt ? "nt   " : "plain", grid, kib, mean * 0.01, mx * 0.01, kib * 1024 / (mean * 10.0), grid * kib * 1024 / (mean * 10.0) / 1000.0);
        }
    return 0;
}
