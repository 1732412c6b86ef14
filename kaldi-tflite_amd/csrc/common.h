// Shared host/device helpers for libktf_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ktf_hip.h"

#define KTF_WAVE 64

void ktf_set_error(const char* fmt, ...);

#define KTF_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            ktf_set_error(__VA_ARGS__);   \
            return KTF_EINVAL;            \
        }                                 \
    } while (0)

#define KTF_CHECK_LAUNCH(name)                                                    \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            ktf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return KTF_ELAUNCH;                                                   \
        }                                                                         \
    } while (0)

// Opt a kernel in to more than 64 KiB of dynamic LDS: once per call site (= kernel instantiation) and device, not per
// launch (a per-launch hipFuncSetAttribute was ~10 host calls per batch-1 extraction).
#define KTF_LDS_ONCE(bytes, ...)                                                                       \
    do {                                                                                               \
        static unsigned long long done_ = 0;                                                           \
        int dev_ = 0;                                                                                  \
        (void)hipGetDevice(&dev_);                                                                     \
        if (!((__atomic_load_n(&done_, __ATOMIC_RELAXED) >> (dev_ & 63)) & 1ull)) {                    \
            (void)hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            (void)__atomic_fetch_or(&done_, 1ull << (dev_ & 63), __ATOMIC_RELAXED);                    \
        }                                                                                              \
    } while (0)

static inline int ktf_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// round-to-nearest-even f32 -> bf16 bits (plain cast keeps NaN a NaN; v_cvt_pk_bf16_f32 at -O3)
__device__ __forceinline__ unsigned short f2bf(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<unsigned short*>(&h);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// Sample g of an utterance's waveform in "frame coordinates" (frame t starts at t*shift - pad_left): outside [0, n) the
// waveform is mirrored with the edge sample repeated, as kaldi_numpy MirrorPad does (frame_extraction.py:28-51).
__device__ __forceinline__ float ktf_wav_sample(const void* wav_b, int i16, int n, int g) {
    if (g < 0) g = -g - 1;
    else if (g >= n) g = 2 * n - 1 - g;
    return i16 ? (float)reinterpret_cast<const short*>(wav_b)[g] : reinterpret_cast<const float*>(wav_b)[g];
}
#endif
