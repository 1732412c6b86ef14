// Error string + version of libktf_hip (the only state the library keeps: thread-local).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ktf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int32_t ktf_version(void) { return 117; /* 0.1.1 + KTF_TDNN_MX_SLAB, KTF_ACT_ELU .. KTF_ACT_SOFTMAX, ktf_activation_f32, KTF_BF16P / KTF_GEMM_BF16X4, ktf_tdnn_split_flat, ktf_tdnn_out_lens; 115: KTF_TDNN_MX_PERSIST, ktf_build_id, ktf_clock_probe; 116: KTF_TDNN_MX_SLAB and KTF_TDNN_MX_PERSIST gone (tools/mx/experiments/); 117: KTF_GEMM_F16 / F16X2 / KTF_F16 gone, ktf_tdnn_split_flat_stats, ktf_stats_finalize_flat, ktf_flat_row_map, row_map argument of ktf_tdnn_split_flat */ }

extern "C" size_t ktf_last_error(char* buf, size_t cap) {
    const size_t n = strlen(g_err);
    if (buf && cap > 0) {
        const size_t c = n < cap - 1 ? n : cap - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return n;
}

// sha256 over the library's sources (csrc/Makefile passes it in): what a measurement kept under profiles/ names as the build it was made on
#ifndef KTF_BUILD_ID
#define KTF_BUILD_ID "unknown"
#endif
extern "C" const char* ktf_build_id(void) { return KTF_BUILD_ID; }

// ------------------------------------------------------------------------------------ shader-clock probe
// One wave that stays on the chip for `us` microseconds next to whatever else runs (it holds no LDS and 8 registers: it fits beside
// a resident GEMM workgroup) and reads the shader clock counter (s_memtime) against the constant 100 MHz counter (s_memrealtime):
// out[0] = shader clocks, out[1] = 100 MHz ticks over the whole stay, out[2] / out[3] = the lowest / highest clock (kHz) seen over
// windows of ~1 ms. The DVFS governor lowers the clock under dense MFMA load by amounts that differ from device to device
// (MI355X_MICROARCH.md, DVFS give-back): a throughput figure is comparable across boxes only next to the clock it was measured at.
__global__ void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long rw = r0, cw = c0, lo = ~0ull, hi = 0, r = r0, c = c0;
    while (r - r0 < ticks) {
        __builtin_amdgcn_s_sleep(127);
        r = __builtin_amdgcn_s_memrealtime();
        c = __builtin_amdgcn_s_memtime();
        if (r - rw >= 100000) {                           // 1 ms
            const unsigned long long khz = (c - cw) * 100000ull / (r - rw);
            lo = khz < lo ? khz : lo;
            hi = khz > hi ? khz : hi;
            rw = r;
            cw = c;
        }
    }
    out[0] = c - c0;
    out[1] = r - r0;
    out[2] = lo == ~0ull ? 0 : lo;
    out[3] = hi;
}

extern "C" int ktf_clock_probe(unsigned long long* out, int64_t us, void* stream) {
    KTF_REQUIRE(out, "ktf_clock_probe: null argument");
    KTF_REQUIRE(us > 0 && us <= 10000000, "ktf_clock_probe: duration outside (0, 10 s]");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, (unsigned long long)us * 100ull);
    KTF_CHECK_LAUNCH("ktf_clock_probe");
    return KTF_OK;
}
