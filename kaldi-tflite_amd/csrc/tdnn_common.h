// Shared by the TDNN GEMM kernel families (tdnn_f32.hip, tdnn_bf16.hip, tdnn_split.hip) and their dispatcher (tdnn_gemm.hip):
// the kernel parameter block, the output-length rule, activations, and the per-family launchers.
//
//   y[b,t,u] = post(act(bias[u] + sum_k sum_d x[b, row(t,k), d] * W[u, k*Dp + d]))
//   row(t,k) = clip(start + t*sub + ctx[k], 0, len_b - 1)
//
// The (T, K*D) im2col matrix of the reference (tf.gather, tdnn.py:258) is never built: the A-tile rows of one K-step all come
// from ONE context offset (Dp is a multiple of the K-step), so staging a tile is a row gather of contiguous 64/128-byte pieces
// straight from the activation matrix, clamped per utterance. M-tiles never straddle utterances (activations are
// utterance-strided), so edge replication needs no row->utterance map.
// Replaces: layers/tdnn/tdnn.py:251-280 (+ keras ReLU, batchnorm.py:78-88) of the reference.
#pragma once
#include <stdlib.h>

#include "common.h"

// Fixed choices of the split-plane kernel (each was an A/B in round 2; DESIGN.md section 5 has the numbers)
#define KTF_X3_Y_NT 1         // the 16-bit activation planes are written with non-temporal stores (the XCD's L2 keeps weights / shared tiles)
#define KTF_X3_WFIRST 1       // the W half of stage 0 is issued before the utterance length is loaded

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bfrag8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float fv4;
typedef __attribute__((ext_vector_type(2))) unsigned uv2;

struct TdnnParams {
    const void* x;
    const int32_t* lens;
    const void* w;
    const void* w_lo;
    const float* bias;
    const float* scale;
    const float* shift;
    void* y;
    const void* x_lo;       // split-bf16 planes (KTF_GEMM_BF16X3 with bf16 x): x = hi plane, x_lo = lo plane
    void* y_lo;             // ... and the same for the output (y = hi plane) when non-null
    int32_t* out_lens;
    int64_t T, ldx, ldy, Tout;
    int32_t units, din_pad, nctx, sub, valid, act, y_dtype, ktot;
    int32_t ctx[16];
    int32_t wtiled;         // KTF_TDNN_W_TILED: W stored as the kernel's LDS images, one contiguous 16 KiB block per (N-tile, K-step)
    int32_t kinter;         // KTF_TDNN_K_INTERLEAVED: K runs (32-wide feature chunk, context, feature) instead of (context, feature)
    int32_t stat_slots;     // fused pooling: 0 = fp64 atomics into (B, 2, units); > 0 = one slot per 128-row block (KTF_TDNN_DET_STATS)
    int32_t y_pair;         // fp32-sized output slots hold the KTF_BF16P pair of the value (y_dtype is KTF_F32 to the store paths)
    const int32_t* row_starts;  // ktf_tdnn_split_flat: (B + 1) exclusive prefix sums of lens (flat row tiling of tdnn_x3s_kernel), else NULL
    const int32_t* row_map;     // ... and optionally ktf_flat_row_map's table: (output row or -1, frame, utterance length, utterance) per flat row
};

// KTF_BF16P: bf16(v) in the low half, bf16(v - bf16(v)) in the high half of a 32-bit slot (tdnn_pair.hip), returned as the float
// with that bit pattern so that the fp32 store paths carry it
__device__ __forceinline__ float ktf_pair(float v) {
    const unsigned hi = f2bf(v);
    const unsigned lo = f2bf(v - bf2f((unsigned short)hi));
    return __uint_as_float(hi | (lo << 16));
}

// Adds (slots == 0) or stores (slots > 0: block `slot` of utterance b is written by exactly one wave) a column's partial sums.
__device__ __forceinline__ void stats_out(double* __restrict__ stats, const TdnnParams& p, int b, int slot, int n, double s, double q) {
    if (p.stat_slots > 0) {
        double* dst = stats + (((int64_t)b * p.stat_slots + slot) * 2) * p.units + n;
        dst[0] = s;
        dst[p.units] = q;
    } else {
        double* dst = stats + ((int64_t)b * 2) * p.units + n;
        atomicAdd(dst, s);
        atomicAdd(dst + p.units, q);
    }
}

__device__ __forceinline__ int tdnn_out_len(int len, const TdnnParams& p, int& start) {
    start = 0;
    int end = len;
    if (p.valid) {
        if (p.ctx[0] < 0) start = -p.ctx[0];
        if (p.ctx[p.nctx - 1] > 0) end = len - p.ctx[p.nctx - 1];
    }
    const int n = end - start;
    return n <= 0 ? 0 : (n + p.sub - 1) / p.sub;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == KTF_ACT_RELU) return v < 0.0f ? 0.0f : v;      // tf.nn.relu propagates NaN (Eigen cwiseMax<PropagateNaN>): the pooled row of an
                                                              // utterance without a frame stays NaN through the layers behind the pooling; fmaxf would return 0
    if (act == KTF_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == KTF_ACT_TANH) return tanhf(v);
    return v;
}

// The rest of tf.keras.activations (TF 2.8: the names layers/tdnn/tdnn.py:117-118 resolves), elementwise ones; the GEMM epilogues
// fuse KTF_ACT_NONE .. KTF_ACT_TANH only, these run as a pass over the layer's output rows (act_rows_kernel, tdnn_gemm.hip).
__device__ __forceinline__ float apply_act_ext(float v, int act) {
    switch (act) {
        case KTF_ACT_ELU: return v > 0.0f ? v : expm1f(v);
        case KTF_ACT_SELU: return 1.05070098735548049342f * (v > 0.0f ? v : 1.67326324235437728481f * expm1f(v));
        case KTF_ACT_SOFTPLUS: return fmaxf(v, 0.0f) + log1pf(expf(-fabsf(v)));
        case KTF_ACT_SOFTSIGN: return v / (fabsf(v) + 1.0f);
        case KTF_ACT_SWISH: return v / (1.0f + expf(-v));
        case KTF_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case KTF_ACT_EXPONENTIAL: return expf(v);
        case KTF_ACT_HARD_SIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.0f), 1.0f);
        default: return apply_act(v, act);
    }
}

// Epilogue for the 32x32 accumulator layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ void store_tile32(const f32x16& acc, const TdnnParams& p, int64_t out_row0, int rows_valid,
                                             int m_base, int n_base, int lane) {
    const int n = n_base + (lane & 31);
    if (n >= p.units) return;
    const float bias = p.bias ? p.bias[n] : 0.0f;
    const float sc = p.scale ? p.scale[n] : 1.0f;
    const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < rows_valid) {
            float v = apply_act(acc[r] + bias, p.act);
            if (p.scale) v = v * sc + sh;
            if (p.y_pair) v = ktf_pair(v);
            const int64_t off = (out_row0 + m) * p.ldy + n;
            if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off] = v;
            else reinterpret_cast<unsigned short*>(p.y)[off] = f2bf(v);
        }
    }
}

// Same tile with the MFMA operands swapped (W block as A, x block as B): the accumulator is the transposed tile,
//   time row = lane&31, unit = (reg&3) + 8*(reg>>2) + 4*(lane>>5),
// so a lane owns four CONSECUTIVE units per register quad and the store is 16 bytes instead of four 4-byte stores (the
// 64 scalar stores per lane of store_tile32 cost the fp32 tile kernel ~20 % of its time). Products commute and the K
// order is unchanged: bit-identical values.
template <int ACT>
__device__ __forceinline__ void store_tile32_t(const f32x16& acc, const TdnnParams& p, int64_t out_row0, int rows_valid,
                                               int m_base, int n_base, int lane) {
    const int m = m_base + (lane & 31);
    if (m >= rows_valid) return;
    const int64_t rowoff = (out_row0 + m) * p.ldy;
    const bool vec_ok = (p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n_base + 8 * q + 4 * (lane >> 5);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool nv = n + e < p.units;
            const float bias = (nv && p.bias) ? p.bias[n + e] : 0.0f;
            v[e] = apply_act(acc[q * 4 + e] + bias, ACT);
            if (p.scale) v[e] = v[e] * (nv ? p.scale[n + e] : 1.0f) + (nv ? p.shift[n + e] : 0.0f);
            if (p.y_pair) v[e] = ktf_pair(v[e]);
        }
        if (p.y_dtype == KTF_F32) {
            float* yp = reinterpret_cast<float*>(p.y) + rowoff + n;
            if (vec_ok && n + 4 <= p.units) {
                *reinterpret_cast<fv4*>(yp) = fv4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.units) yp[e] = v[e];
            }
        } else {
            unsigned short* yp = reinterpret_cast<unsigned short*>(p.y) + rowoff + n;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.units) yp[e] = f2bf(v[e]);
        }
    }
}


// name of the kernel family the calling thread's last ktf_tdnn* call launched (ktf_tdnn_last_kernel; dispatch tests)
extern thread_local const char* g_ktf_last_kernel;
#define KTF_NOTE_KERNEL(name) (g_ktf_last_kernel = (name))

// per-family launchers (validation of the family's own constraints + launch); `p` is filled by tdnn_gemm.hip
int tdnn_launch_f32(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, hipStream_t st);
int tdnn_launch_x4(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, double* stats, hipStream_t st);   // tdnn_pair.hip
int tdnn_launch_16(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, int64_t ldy, double* stats_sums, hipStream_t st);
int tdnn_launch_split(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, int64_t ldy, bool split_in, double* stats_sums,
                      hipStream_t st);
