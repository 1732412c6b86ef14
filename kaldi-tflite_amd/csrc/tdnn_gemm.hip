// TDNN layer = implicit-im2col GEMM on the gfx950 matrix cores, with the layer's bias,
// activation and the following BatchNorm (as a per-unit affine) fused into the epilogue.
//
//   y[b,t,u] = post(act(bias[u] + sum_k sum_d x[b, row(t,k), d] * W[u, k*Dp + d]))
//   row(t,k) = clip(start + t*sub + ctx[k], 0, len_b - 1)
//
// The (T, K*D) im2col matrix of the reference (tf.gather, tdnn.py:258) is never built: the
// A-tile rows of one K-step all come from ONE context offset (Dp is a multiple of the K-step),
// so staging a tile is a row gather of contiguous 64/128-byte pieces straight from the
// activation matrix, clamped per utterance. M-tiles never straddle utterances (activations
// are utterance-strided), so edge replication needs no row->utterance map.
//
// Three arithmetic modes (KtfTdnnDesc.gemm):
//   F32    v_mfma_f32_32x32x2_f32  — exact fp32 products / fp32 accumulate (bit-identical to an
//          fmaf chain); the parity path.
//   BF16   v_mfma_f32_32x32x16_bf16 — bf16 operands, fp32 accumulate; the throughput path.
//   BF16X3 x = hi+lo, w = hi+lo (bf16 pairs), acc += hi*hi + lo*hi + hi*lo — ~16 mantissa bits
//          at 3 MFMA passes.
//
// Replaces: layers/tdnn/tdnn.py:251-280 (+ keras ReLU, batchnorm.py:78-88) of the reference.
#include <stdlib.h>

#include "common.h"

// Kernel-selection knobs and the per-tile stamp buffer exist in probe builds only (-DKTF_TILE_PROBE, tools/tile_probe.py):
// the product library reads no environment variable and keeps no state.
#ifdef KTF_TILE_PROBE
#define KTF_KNOB(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
static long long* g_probe_buf = nullptr;
extern "C" void ktf_probe_set_buffer(void* p) { g_probe_buf = (long long*)p; }
#define KTF_PROBE_BUF g_probe_buf
#else
#define KTF_KNOB(name, dflt) (dflt)
#define KTF_PROBE_BUF ((long long*)nullptr)
#endif
#ifndef KTF_X3_A_AUX
#define KTF_X3_A_AUX 0        // cache policy of the steady-state operand DMAs of the split-plane kernel (0 default, 2 = nt, 1 = sc0, 16 = sc1)
#endif
#ifndef KTF_X3_W_AUX
#define KTF_X3_W_AUX 0
#endif
#ifndef KTF_X3_Y_NT
#define KTF_X3_Y_NT 1         // 1: the 16-bit activation planes are written with non-temporal stores (0: A/B)
#endif
#ifndef KTF_X1_STAGES
#define KTF_X1_STAGES 3       // one-pass form: 32 KiB stages in the LDS ring (2, 3 or 4). Its K-steps are bound by the operand stream, so
                              // a second K-step of DMAs in flight pays: tdnn4 0.73 -> 0.67 ms, tdnn5 1.80 -> 1.78 ms, +1.2 % on the step; a third
                              // does not (4 stages: -0.5 %)
#endif
#ifndef KTF_X2_PIPE
#define KTF_X2_PIPE 1         // K-loop of the two-pass half form: 1 = in-phase hand-scheduled step (default), 2 = ping-pong wave halves.
                              // Measured equal (97.6-98.3 k vs 96.3-97.7 k x-vectors/s on one box), and the ping-pong stamps
                              // (tools/pp_seg_probe.py, K-step 10 of tdnn2) say why: MFMA segment 1028 cycles as paced, DMA issue
                              // 470-560, fragment reads 250-340 -- and then 920-980 cycles at the counted vmcnt wait in front of the
                              // barrier, in BOTH groups: the operand DMAs issued one K-step (2900 cycles) earlier have not landed.
                              // With s_sleep in place of the MFMAs (same segment length) that wait is 250-300 cycles; every tile
                              // reading the same (L2-hot) activations changes it by 4 %. Timing-only ablations of the whole step
                              // (tdnn2 + tdnn3, ms): all 4.98, no DMA 3.76, no fragment reads 4.56, no MFMA 2.29, MFMA alone 3.62,
                              // DMA alone 2.29 (64 GB/s per CU). An LDS-DMA stream that runs at 64 GB/s per CU beside idle matrix
                              // pipes delivers ~34 GB/s beside busy ones: the K-step is bound by that, not by how the waves
                              // interleave their instructions.
                              // Followed up and not kept: (a) without the steady-state activation DMAs (timing-only) the step is
                              // 9 % shorter, without the weight-residual DMAs 7 %; (b) a shared activation window for the
                              // multi-context layers (ONE 272-row window per 32-feature chunk read at row offset 8 + ctx by its
                              // contexts' K-steps: operand DMA bytes per step 48 -> 37.7 KiB, bit-identical results) gave 1.5 % on
                              // those layers, 0.4 % on the step, the same with the window fetched one or two chunks ahead: it
                              // removes re-reads that hit in L2, not the first touch that comes from HBM, and it is the latter's
                              // traffic the remaining operand stream competes with; (c) non-temporal operand loads cost 3-6 %;
                              // non-temporal STORES of the activation plane are kept (+1.2 %).
#endif
#ifndef KTF_X3_PRIO
#define KTF_X3_PRIO 0
#endif
#ifndef KTF_X3_WFIRST
#define KTF_X3_WFIRST 1       // split-plane kernel: the W half of stage 0 is issued before the utterance length is loaded
#endif
#ifndef KTF_X3S_DEFAULT
#define KTF_X3S_DEFAULT 2     // split-bf16 planes: 16x16x32 kernel for every layer (measured 62.3 k vs 58.6 k x-vectors/s with 1 = pooling layer only)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bfrag8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
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
    int32_t xchunk, ychunk; // KTF_TDNN_X_CHUNKED / KTF_TDNN_Y_CHUNKED: 16-bit activations stored (utterance, 32-feature chunk, row, 32)
    int32_t wtiled;         // KTF_TDNN_W_TILED: W stored as the kernel's LDS images, one contiguous 16 KiB block per (N-tile, K-step)
    int32_t kinter;         // KTF_TDNN_K_INTERLEAVED: K runs (32-wide feature chunk, context, feature) instead of (context, feature)
#ifdef KTF_TILE_PROBE
    long long* probe;       // per-tile s_memrealtime stamps (probe builds)
#endif
    int32_t stat_slots;     // fused pooling: 0 = fp64 atomics into (B, 2, units); > 0 = one slot per 128-row block (KTF_TDNN_DET_STATS)
    int32_t lo_steps;       // F16X2: K-steps [0, lo_steps) run two passes, the rest one (KTF_TDNN_LO_PREFIX); >= ktot / 32: all of them
};

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
    if (act == KTF_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == KTF_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == KTF_ACT_TANH) return tanhf(v);
    return v;
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

// ------------------------------------------------------------------------------------ F32
// (64*MT) x (64*MT) block tile, K-step 16, 4 waves as 2x2, each wave MT x MT MFMA 32x32 tiles. MT = 2 (128x128) is the
// throughput shape; MT = 1 (64x64) is used when the 128-tiles would fill fewer workgroups than the chip has CUs (one
// utterance: M = 998 -> 32 workgroups; tdnn6: one row per utterance): four times the workgroups, a quarter of the
// serial MFMA chain per wave.
// K-step BK: 16 for the big tile; 32 for the small one, whose MFMA time per K-step is too short to cover a global load.
template <int MT, int BK>
__global__ __launch_bounds__(256) void tdnn_f32_kernel(TdnnParams p) {
    constexpr int BM = 64 * MT, BN = 64 * MT;
    constexpr int F32_BK = BK, F32_PITCH = BK + 1;
    constexpr int C4 = BK / 4;                       // float4 per staged row
    constexpr int NLD = BM * C4 / 256;               // float4 per thread and operand
    __shared__ float As[2][BM * F32_PITCH];
    __shared__ float Bs[2][BN * F32_PITCH];
    const int b = blockIdx.z;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = blockIdx.y * BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const float* xb = reinterpret_cast<const float*>(p.x) + (int64_t)b * p.T * p.ldx;
    const float* wb = reinterpret_cast<const float*>(p.w);

    // staging map: float4 q = i*256 + tid of the (BM x BK) slice -> row q / C4, column 4*(q % C4)
    int ld_row[NLD], ld_col[NLD], a_t[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int q = i * 256 + tid;
        ld_row[i] = q / C4;
        ld_col[i] = (q % C4) * 4;
        a_t[i] = start + (t0 + ld_row[i]) * p.sub;
    }

    f32x16 acc[MT][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / F32_BK;
    const int steps_per_ctx = p.din_pad / F32_BK;
    float4 ra[NLD], rb[NLD];

    auto load_global = [&](int ks) {
        const int c = ks / steps_per_ctx;
        const int d0 = (ks - c * steps_per_ctx) * F32_BK;
        const int off = p.ctx[c];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int r = a_t[i] + off;
            r = r < 0 ? 0 : (r > len - 1 ? len - 1 : r);
            ra[i] = *reinterpret_cast<const float4*>(xb + (int64_t)r * p.ldx + d0 + ld_col[i]);
            rb[i] = *reinterpret_cast<const float4*>(wb + (int64_t)(n0 + ld_row[i]) * p.ktot + ks * F32_BK + ld_col[i]);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float* a = &As[buf][ld_row[i] * F32_PITCH + ld_col[i]];
            a[0] = ra[i].x; a[1] = ra[i].y; a[2] = ra[i].z; a[3] = ra[i].w;
            float* bb = &Bs[buf][ld_row[i] * F32_PITCH + ld_col[i]];
            bb[0] = rb[i].x; bb[1] = rb[i].y; bb[2] = rb[i].z; bb[3] = rb[i].w;
        }
    };

    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_global(ks + 1);
        const float* a_base = &As[buf][(wm * 32 * MT + (lane & 31)) * F32_PITCH + (lane >> 5)];
        const float* b_base = &Bs[buf][(wn * 32 * MT + (lane & 31)) * F32_PITCH + (lane >> 5)];
#pragma unroll
        for (int kk = 0; kk < F32_BK; kk += 2) {
            float av[MT], bv[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                av[i] = a_base[i * 32 * F32_PITCH + kk];
                bv[i] = b_base[i * 32 * F32_PITCH + kk];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j], av[i], acc[i][j], 0, 0, 0);   // transposed tile
        }
        if (ks + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    // the activation is a compile-time constant inside each copy: with the runtime switch inlined per value the epilogue
    // was ~10k instructions of branches (sigmoid / tanh bodies 64 times over) and took 60-130 us per tile -- longer than
    // the K-loop of the K = 512 layers (in-kernel s_memrealtime stamps)
#define F32_EPILOGUE(A)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                     \
        _Pragma("unroll") for (int j = 0; j < MT; ++j)                                                                 \
            store_tile32_t<A>(acc[i][j], p, out_row0, rows_valid, wm * 32 * MT + i * 32, n0 + wn * 32 * MT + j * 32, lane);
    if (p.act == KTF_ACT_NONE) { F32_EPILOGUE(KTF_ACT_NONE) }
    else if (p.act == KTF_ACT_RELU) { F32_EPILOGUE(KTF_ACT_RELU) }
    else if (p.act == KTF_ACT_SIGMOID) { F32_EPILOGUE(KTF_ACT_SIGMOID) }
    else { F32_EPILOGUE(KTF_ACT_TANH) }
#undef F32_EPILOGUE
}

// ------------------------------------------------------------------------------------ BF16 / BF16X3
// 128x128 block tile, K-step BK (bf16), 4 waves as 2x2, each wave 64x64 = 2x2 tiles of 32x32x16 MFMA.
// LDS rows are padded by 16 B so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots.
#define BF_BM 128
#define BF_BN 128

template <int BK>
struct BfCfg {
    static constexpr int PITCH = BK + 8;                    // bf16 elements per LDS row
    static constexpr int CHUNKS = BK / 8;                   // 16-B chunks per row
    static constexpr int PER_THREAD = (128 * CHUNKS) / 256; // chunks each thread stages per operand
};

__device__ __forceinline__ u32x4 pack_bf16x8(const fv4& lo, const fv4& hi) {
    u32x4 r;
    r.x = (unsigned)f2bf(lo.x) | ((unsigned)f2bf(lo.y) << 16);
    r.y = (unsigned)f2bf(lo.z) | ((unsigned)f2bf(lo.w) << 16);
    r.z = (unsigned)f2bf(hi.x) | ((unsigned)f2bf(hi.y) << 16);
    r.w = (unsigned)f2bf(hi.z) | ((unsigned)f2bf(hi.w) << 16);
    return r;
}
__device__ __forceinline__ fv4 bf_residual(const fv4& v, unsigned p01, unsigned p23) {
    fv4 r;
    r.x = v.x - bf2f((unsigned short)(p01 & 0xffff)); r.y = v.y - bf2f((unsigned short)(p01 >> 16));
    r.z = v.z - bf2f((unsigned short)(p23 & 0xffff)); r.w = v.w - bf2f((unsigned short)(p23 >> 16));
    return r;
}

// XF32: activations are fp32 in memory (converted while staging); X3: split-bf16 3-pass mode (needs XF32).
template <int BK, bool XF32, bool X3>
__global__ __launch_bounds__(256) void tdnn_bf16_kernel(TdnnParams p) {
    using C = BfCfg<BK>;
    constexpr int NBUF_A = X3 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    // layout: [stage 2][A hi (,A lo)][B hi (,B lo)] each 128 x PITCH
    constexpr int TILE = 128 * C::PITCH;
    constexpr int STAGE = TILE * 2 * NBUF_A;

    const int b = blockIdx.z;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = blockIdx.y * BF_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = blockIdx.x * BF_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int64_t xbase = (int64_t)b * p.T * p.ldx;
    const unsigned short* wb = reinterpret_cast<const unsigned short*>(p.w);
    const unsigned short* wlo = reinterpret_cast<const unsigned short*>(p.w_lo);

    int ld_row[C::PER_THREAD], ld_chunk[C::PER_THREAD], a_t[C::PER_THREAD];
#pragma unroll
    for (int i = 0; i < C::PER_THREAD; ++i) {
        const int id = tid + 256 * i;
        ld_row[i] = id / C::CHUNKS;
        ld_chunk[i] = id % C::CHUNKS;
        a_t[i] = start + (t0 + ld_row[i]) * p.sub;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / BK;
    const int steps_per_ctx = p.din_pad / BK;
    u32x4 ra[C::PER_THREAD], ralo[C::PER_THREAD], rb[C::PER_THREAD], rblo[C::PER_THREAD];

#define BF_LOAD_GLOBAL(KS)                                                                                        \
    {                                                                                                             \
        const int ks_ = (KS);                                                                                     \
        const int c_ = ks_ / steps_per_ctx;                                                                       \
        const int d0_ = (ks_ - c_ * steps_per_ctx) * BK;                                                          \
        const int off_ = p.ctx[c_];                                                                               \
        _Pragma("unroll") for (int i = 0; i < C::PER_THREAD; ++i) {                                               \
            int r = a_t[i] + off_;                                                                                \
            r = r < 0 ? 0 : (r > len - 1 ? len - 1 : r);                                                          \
            const int64_t e = xbase + (int64_t)r * p.ldx + d0_ + ld_chunk[i] * 8;                                 \
            if (XF32) {                                                                                           \
                const fv4* src = reinterpret_cast<const fv4*>(reinterpret_cast<const float*>(p.x) + e);           \
                const fv4 v0 = src[0], v1 = src[1];                                                               \
                ra[i] = pack_bf16x8(v0, v1);                                                                      \
                if (X3) ralo[i] = pack_bf16x8(bf_residual(v0, ra[i].x, ra[i].y), bf_residual(v1, ra[i].z, ra[i].w)); \
            } else {                                                                                              \
                ra[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.x) + e);        \
            }                                                                                                     \
            const int64_t we = (int64_t)(n0 + ld_row[i]) * p.ktot + (int64_t)ks_ * BK + ld_chunk[i] * 8;          \
            rb[i] = *reinterpret_cast<const u32x4*>(wb + we);                                                     \
            if (X3) rblo[i] = *reinterpret_cast<const u32x4*>(wlo + we);                                          \
        }                                                                                                         \
    }
#define BF_STORE_LDS(STG)                                                                      \
    {                                                                                          \
        unsigned short* base = smem + (STG) * STAGE;                                           \
        _Pragma("unroll") for (int i = 0; i < C::PER_THREAD; ++i) {                            \
            const int o = ld_row[i] * C::PITCH + ld_chunk[i] * 8;                              \
            *reinterpret_cast<u32x4*>(base + o) = ra[i];                                       \
            if (X3) *reinterpret_cast<u32x4*>(base + TILE + o) = ralo[i];                      \
            *reinterpret_cast<u32x4*>(base + NBUF_A * TILE + o) = rb[i];                       \
            if (X3) *reinterpret_cast<u32x4*>(base + NBUF_A * TILE + TILE + o) = rblo[i];      \
        }                                                                                      \
    }

    BF_LOAD_GLOBAL(0);
    BF_STORE_LDS(0);
    __syncthreads();
    // fragment base offsets: lane (r = lane&31, h = lane>>5) reads row r, k = 16*kstep + 8*h .. +7
    const int a_off = (wm * 64 + (lane & 31)) * C::PITCH + (lane >> 5) * 8;
    const int b_off = (wn * 64 + (lane & 31)) * C::PITCH + (lane >> 5) * 8;
    for (int ks = 0; ks < nk; ++ks) {
        const int stage = ks & 1;
        if (ks + 1 < nk) BF_LOAD_GLOBAL(ks + 1);
        const unsigned short* sa = smem + stage * STAGE;
        const unsigned short* sb = sa + NBUF_A * TILE;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            bfrag8 a[2], bq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bfrag8*>(sa + a_off + i * 32 * C::PITCH + kk);
                bq[i] = *reinterpret_cast<const bfrag8*>(sb + b_off + i * 32 * C::PITCH + kk);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], acc[i][j], 0, 0, 0);
            if (X3) {
                bfrag8 al[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    al[i] = *reinterpret_cast<const bfrag8*>(sa + TILE + a_off + i * 32 * C::PITCH + kk);
                    bl[i] = *reinterpret_cast<const bfrag8*>(sb + TILE + b_off + i * 32 * C::PITCH + kk);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bq[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (ks + 1 < nk) BF_STORE_LDS(stage ^ 1);
        __syncthreads();
    }

    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile32(acc[i][j], p, out_row0, rows_valid, wm * 64 + i * 32, n0 + wn * 64 + j * 32, lane);
}

// ------------------------------------------------------------------------------------ BF16, direct-to-LDS staging
// The throughput kernel for bf16 activations: 128x128 tile, K-step 64, both operands staged with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write). The LDS image is lane-linear ([row][64] bf16, 128-B rows),
// so bank conflicts of the ds_read_b128 fragment reads are removed by permuting the 16-B chunks of each row on the
// SOURCE address (chunk' = chunk ^ ((row>>1)&7)) and applying the same involution on the read address.
// 1-D grid, XCD-aware: block id -> (xcd = id % 8, slot = id / 8); an XCD walks its own M-tiles and runs all N-tiles of
// one M-tile back to back, so the gathered activation rows are fetched into that XCD's L2 once.
// The epilogue stages the fp32 accumulators through LDS and writes whole 256-B row segments.
// cache policy bits of the operand DMAs (aux of global_load_lds: 1 = sc0, 2 = nt, 16 = sc1); A = activations, W = weights.
// Measured (tools/gemm_layers.py): nt on the activations -10..-20 %, nt on the weights -10..-40 %, sc0 no change: both
// streams live on L2 hits (other N-tiles / context offsets re-read the activations, every CU re-reads the weights).
#ifndef KTF_AUX_A
#define KTF_AUX_A 0
#endif
#ifndef KTF_AUX_W
#define KTF_AUX_W 0
#endif
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;

#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_TILE_BYTES (128 * G_BK * 2)          // one operand tile: 16 KiB
#define G_STAGE_BYTES (2 * G_TILE_BYTES)       // A + B
#define G_EPI_PITCH 132                         // floats per staged output row
#define G_LDS_BYTES (128 * G_EPI_PITCH * 4)    // 67,584 B >= 2 stages (65,536 B)

__global__ __launch_bounds__(256) void tdnn_bf16g_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int g = (slot / ntiles) * 8 + xcd;     // global M-tile index
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = g / mtiles, mt = g - b * mtiles;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && nt == 0 && mt == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = mt * G_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = nt * G_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const unsigned short* xb = reinterpret_cast<const unsigned short*>(p.x) + (int64_t)b * p.T * p.ldx;
    const unsigned short* wb = reinterpret_cast<const unsigned short*>(p.w);

    // staging map: chunk q = i*256 + tid -> row q/8, LDS position q%8, global chunk (q%8) ^ ((row>>1)&7)
    int a_t[4];
    int src_chunk[4];
    const unsigned short* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid;
        const int row = q >> 3;
        src_chunk[i] = ((q & 7) ^ ((row >> 1) & 7)) * 8;
        a_t[i] = start + (t0 + row) * p.sub;
        wrow[i] = wb + (int64_t)(n0 + row) * p.ktot + src_chunk[i];
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / G_BK;
    const int steps_per_ctx = p.din_pad / G_BK;

#define G_STAGE(STG, KS)                                                                                              \
    {                                                                                                                 \
        const int ks_ = (KS);                                                                                         \
        const int c_ = ks_ / steps_per_ctx;                                                                           \
        const int d0_ = (ks_ - c_ * steps_per_ctx) * G_BK;                                                            \
        const int off_ = p.ctx[c_];                                                                                   \
        unsigned char* sa_ = gsm + (STG) * G_STAGE_BYTES + wave * 1024;                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
            int r = a_t[i] + off_;                                                                                    \
            r = r < 0 ? 0 : (r > len - 1 ? len - 1 : r);                                                              \
            const unsigned short* ga = xb + (int64_t)r * p.ldx + d0_ + src_chunk[i];                                  \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)ga, (lds_ptr_t*)(sa_ + i * 4096), 16, 0, 0);                  \
            const unsigned short* gb = wrow[i] + (int64_t)ks_ * G_BK;                                                 \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)gb, (lds_ptr_t*)(sa_ + G_TILE_BYTES + i * 4096), 16, 0, 0);  \
        }                                                                                                             \
    }

    G_STAGE(0, 0);
    __syncthreads();
    // fragment addressing: lane (r = lane&31, h = lane>>5) reads row R, k = kk + 8h .. +7  ->  chunk (kk/8 + h) ^ ((R>>1)&7)
    const int rsw = ((lane & 31) >> 1) & 7;
    const int a_row_off = (wm * 64 + (lane & 31)) * 128;   // bytes
    const int b_row_off = (wn * 64 + (lane & 31)) * 128;
    const int hsel = lane >> 5;
    for (int ks = 0; ks < nk; ++ks) {
        const int stage = ks & 1;
        if (ks + 1 < nk) G_STAGE(stage ^ 1, ks + 1);
        const unsigned char* sa = gsm + stage * G_STAGE_BYTES;
        const unsigned char* sb = sa + G_TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < G_BK / 16; ++kk) {
            const int coff = (((kk * 2 + hsel) ^ rsw) << 4);
            bfrag8 a[2], bq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i * 32 * 128 + coff);
                bq[i] = *reinterpret_cast<const bfrag8*>(sb + b_row_off + i * 32 * 128 + coff);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#undef G_STAGE

    // ---- epilogue: bias / activation / BN affine on the accumulators, stage fp32 tile in LDS, coalesced row stores
    float* et = reinterpret_cast<float*>(gsm);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nl = wn * 64 + j * 32 + (lane & 31);
        const int n = n0 + nl;
        const bool nv = n < p.units;
        const float bias = (nv && p.bias) ? p.bias[n] : 0.0f;
        const float sc = (nv && p.scale) ? p.scale[n] : 1.0f;
        const float sh = (nv && p.shift) ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = apply_act(acc[i][j][r] + bias, p.act);
                if (p.scale) v = v * sc + sh;
                et[m * G_EPI_PITCH + nl] = v;
            }
        }
    }
    __syncthreads();
    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    const int cl = (tid & 15) * 4;                 // 4 columns at cl and 4 at 64 + cl
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int m = pass * 16 + (tid >> 4);
        if (m >= rows_valid) continue;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int nl = half * 64 + cl;
            const int n = n0 + nl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(et + m * G_EPI_PITCH + nl);
            const int64_t off = (out_row0 + m) * p.ldy + n;
            if (n + 4 <= p.units) {
                if (p.y_dtype == KTF_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + off) = v;
                } else {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
                    pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + off) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < p.units) {
                        if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off + e] = v[e];
                        else reinterpret_cast<unsigned short*>(p.y)[off + e] = f2bf(v[e]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ BF16, 256x256 tile, 4-deep LDS ring
// The large-layer throughput kernel. One workgroup = 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA 32x32 tiles) owns a
// 256 x 256 output tile: 32 B of staged operand per MFMA-cycle-pair instead of 64 (the 128x128 kernel is L2->LDS bound).
// Operands are staged with global_load_lds_dwordx4 into a ring of four 32 KiB stages (K-step 32: 64-B rows, chunk
// permutation chunk ^ ((row>>2)&3) on the source, same involution on the read). Loads run THREE K-steps ahead and stay
// in flight across the single raw s_barrier per K-step: the wait before the barrier is a counted s_waitcnt vmcnt(8|4|0)
// (4 DMA instructions per thread per stage), never a drain.
#define R_BM 256
#define R_BN 256
#define R_BK 32
#define R_NSTAGE 4
#define R_TILE_BYTES (256 * R_BK * 2)           // 16 KiB per operand
#define R_STAGE_BYTES (2 * R_TILE_BYTES)        // 32 KiB
#define R_EPI_PITCH 260
#define R_LDS_BYTES (R_NSTAGE * R_STAGE_BYTES)  // 131,072 B (epilogue staging needs 64*260*4 = 66,560 B)

// Epilogue shared by the 256x256 kernels: bias -> activation -> BatchNorm affine on the 4x2 accumulator tiles of each wave,
// then either (STATS) fp64 column sums / sums of squares into stats[b][0|1][unit], or four passes of LDS-staged,
// fully coalesced row stores (one 256-column row per wave-instruction).
template <int ACT, bool STATS>
__device__ __forceinline__ void ring_epilogue(f32x16 (&acc)[4][2], const TdnnParams& p, double* __restrict__ stats,
                                              unsigned char* rsm, int b, int t0, int n0, int out_len, int wm, int wn,
                                              int wave, int lane) {
    // ---- epilogue: four passes of 64 staged rows (wave (wm, wn) contributes its 32 x 64 block of pass i)
    float* et = reinterpret_cast<float*>(rsm);
    float bias[2], sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        const bool nv = n < p.units;
        bias[j] = (nv && p.bias) ? p.bias[n] : 0.0f;
        sc[j] = (nv && p.scale) ? p.scale[n] : 1.0f;
        sh[j] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }
    const int rows_valid = out_len - t0;
    if (STATS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r] + bias[j];
                    if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                    v = v * sc[j] + sh[j];
                    if (m < rows_valid) {
                        s += (double)v;
                        q += (double)v * (double)v;
                    }
                }
            }
            s += __shfl_xor(s, 32, 64);      // the two half-waves hold the same column
            q += __shfl_xor(q, 32, 64);
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (lane < 32 && n < p.units) stats_out(stats, p, b, (t0 >> 7) + wm, n, s, q);
        }
        return;
    }
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    const int nl = lane * 4;                      // this lane's 4 columns of the 256-wide staged row
    const int n = n0 + nl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r] + bias[j];
                if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                v = v * sc[j] + sh[j];
                et[srow * R_EPI_PITCH + col] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            const int srow = sp * 8 + wave;          // one staged row per wave: 256 contiguous columns
            const int m = (srow >> 5) * 128 + i * 32 + (srow & 31);
            if (m < rows_valid) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + nl);
                const int64_t off = (out_row0 + m) * p.ldy + n;
                if (n + 4 <= p.units) {
                    if (p.y_dtype == KTF_F32) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + off) = v;
                    } else {
                        const unsigned short h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
                        uint2 pk;
                        pk.x = (unsigned)h0 | ((unsigned)h1 << 16);
                        pk.y = (unsigned)h2 | ((unsigned)h3 << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + off) = pk;
                        if (p.y_lo) {            // split-bf16 output: the residual plane, the next layer's lo operand
                            uint2 pl;
                            pl.x = (unsigned)f2bf(v.x - bf2f(h0)) | ((unsigned)f2bf(v.y - bf2f(h1)) << 16);
                            pl.y = (unsigned)f2bf(v.z - bf2f(h2)) | ((unsigned)f2bf(v.w - bf2f(h3)) << 16);
                            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y_lo) + off) = pl;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.units) {
                            if (p.y_dtype == KTF_F32) {
                                reinterpret_cast<float*>(p.y)[off + e] = v[e];
                            } else {
                                const unsigned short h = f2bf(v[e]);
                                reinterpret_cast<unsigned short*>(p.y)[off + e] = h;
                                if (p.y_lo) reinterpret_cast<unsigned short*>(p.y_lo)[off + e] = f2bf(v[e] - bf2f(h));
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// STATS: instead of storing y, the epilogue adds every column's sum and sum of squares over the tile's valid rows (fp64)
// into stats[b][0|1][unit] — statistics pooling fused into the producing GEMM, the (B,T,units) activation never exists.
template <int ACT, bool STATS>
__global__ __launch_bounds__(512) void tdnn_bf16r_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
                                                         double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = g / mtiles, mt = g - b * mtiles;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && nt == 0 && mt == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = mt * R_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = nt * R_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay in SGPRs
    const int wm = wave >> 2, wn = wave & 3;

    // Uniform 64-bit bases + per-lane 32-bit byte offsets: every DMA address is base(SGPR) + offset(VGPR), so the K-loop
    // carries no 64-bit vector arithmetic (an utterance's activations and a layer's weights are both < 4 GiB).
    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * 2;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const unsigned ldxb = (unsigned)p.ldx * 2u;

    // staging map: chunk q = i*512 + tid -> row q/4, LDS position q%4, global chunk (q%4) ^ ((row>>2)&3)
    int a_t[2];
    unsigned a_cb[2], w_ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        const int row = q >> 2;
        const unsigned chunk = (unsigned)(((q & 3) ^ ((row >> 2) & 3)) * 16);   // bytes
        a_cb[i] = chunk;
        a_t[i] = start + (t0 + row) * p.sub;
        w_ob[i] = (unsigned)(n0 + row) * (unsigned)p.ktot * 2u + chunk;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / R_BK;
    const int lenm1 = len - 1;

    // iterator over the stage being issued: K-step index, context offset of its rows, byte offset inside the context
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 2;
#define R_DMA_A(i)                                                                                                     \
    {                                                                                                                  \
        int r_ = a_t[i] + is_off;                                                                                      \
        r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                                   \
        const unsigned vo_ = (unsigned)r_ * ldxb + a_cb[i] + (unsigned)is_db;                                          \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + vo_),                                                       \
            (lds_ptr_t*)(rsm + (is_ks & (R_NSTAGE - 1)) * R_STAGE_BYTES + wave * 1024 + (i) * 8192), 16, 0, KTF_AUX_A);\
    }
#define R_DMA_B(i)                                                                                                     \
    {                                                                                                                  \
        const unsigned vo_ = w_ob[i] + (unsigned)(is_ks * (R_BK * 2));                                                 \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + vo_),                                                       \
            (lds_ptr_t*)(rsm + (is_ks & (R_NSTAGE - 1)) * R_STAGE_BYTES + R_TILE_BYTES + wave * 1024 + (i) * 8192),    \
            16, 0, KTF_AUX_W);                                                                                            \
    }
#define R_ADVANCE()                                                                                                    \
    {                                                                                                                  \
        ++is_ks;                                                                                                       \
        is_db += R_BK * 2;                                                                                             \
        if (is_db == dpad_b) {                                                                                         \
            is_db = 0;                                                                                                 \
            ++is_c;                                                                                                    \
            is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                                \
        }                                                                                                              \
    }

    // prologue: three stages in flight
    for (int s_ = 0; s_ < 3 && s_ < nk; ++s_) {
        R_DMA_A(0) R_DMA_B(0) R_DMA_A(1) R_DMA_B(1)
        R_ADVANCE()
    }

    const int rsw = ((lane & 31) >> 2) & 3;
    const int a_row_off = (wm * 128 + (lane & 31)) * 64;   // bytes (64-B rows)
    const int b_row_off = (wn * 64 + (lane & 31)) * 64;
    const int hsel = lane >> 5;
    const int coff0 = ((hsel ^ rsw) << 4), coff1 = (((2 + hsel) ^ rsw) << 4);
    // Software pipeline: the barrier of K-step ks certifies stages ks AND ks+1 (one stage = 4 DMA instructions per thread
    // stays in flight), so the first-half fragments of stage ks+1 are read during the MFMAs of stage ks and the matrix
    // pipe restarts right after the next barrier instead of waiting for an LDS read burst of all 8 lock-stepped waves.
    bfrag8 a0[4], b0[2];     // fragments of (current stage, k-half 0)
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // every wave is past the barrier, i.e. has finished reading stage ks-1: that buffer is refilled with stage ks+3
        // (the iterator's stage); its four DMA instructions are spread between the MFMA groups
        const bool refill = is_ks < nk;
        const unsigned char* sa = rsm + (ks & (R_NSTAGE - 1)) * R_STAGE_BYTES;
        const unsigned char* sb = sa + R_TILE_BYTES;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[i] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i * 32 * 64 + coff0);
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[j] = *reinterpret_cast<const bfrag8*>(sb + b_row_off + j * 32 * 64 + coff0);
        }
        bfrag8 a1[4], b1[2];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = half * 2; i < half * 2 + 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keep the MFMA group ahead of the LDS reads / DMA that follow it
            if (half == 0) {
                // second-half fragments of this stage: issued behind the first MFMA group so their latency is covered
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i * 32 * 64 + coff1);
#pragma unroll
                for (int j = 0; j < 2; ++j) b1[j] = *reinterpret_cast<const bfrag8*>(sb + b_row_off + j * 32 * 64 + coff1);
            }
            if (refill) {
                if (half == 0) R_DMA_A(0) else R_DMA_B(0)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = half * 2; i < half * 2 + 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (half == 0 && ks + 1 < nk) {
                // pre-read the first-half fragments of stage ks+1 (landed and visible since this K-step's barrier); all
                // MFMAs that consume the old a0/b0 have been issued
                const unsigned char* san = rsm + ((ks + 1) & (R_NSTAGE - 1)) * R_STAGE_BYTES;
                const unsigned char* sbn = san + R_TILE_BYTES;
#pragma unroll
                for (int i = 0; i < 4; ++i) a0[i] = *reinterpret_cast<const bfrag8*>(san + a_row_off + i * 32 * 64 + coff0);
#pragma unroll
                for (int j = 0; j < 2; ++j) b0[j] = *reinterpret_cast<const bfrag8*>(sbn + b_row_off + j * 32 * 64 + coff0);
            }
            if (refill) {
                if (half == 0) R_DMA_A(1) else R_DMA_B(1)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (is_ks < nk) R_ADVANCE()
    }
#undef R_DMA_A
#undef R_DMA_B
#undef R_ADVANCE
#define R_STAGE
#undef R_STAGE
    __syncthreads();   // all fragment reads done before the LDS is reused by the epilogue

    ring_epilogue<ACT, STATS>(acc, p, stats, rsm, b, t0, n0, out_len, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------ BF16, 256x256 tile, 16x16x32 MFMA
// Same ring / DMA / tile order as tdnn_bf16r_kernel, but the wave's 128 x 64 block is 8 x 4 tiles of
// v_mfma_f32_16x16x32_bf16: one MFMA consumes the whole 32-deep K-step, and the chip holds a higher clock on this
// shape under load (MI355X_MICROARCH.md, DVFS item 7). Fragment lane map: row = lane&15, 16-B chunk = lane>>4, so the
// conflict-free chunk permutation is c ^ ((4 - (row>>2)) & 3) (each ds_read_b128 lane group then covers all 16 slots).
typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(8))) _Float16 hfrag8;
// one 16x16x32 MFMA on 16-bit fragments held as raw 16 bytes: bf16 or (F16) IEEE half operands
template <bool F16>
__device__ __forceinline__ f32x4v mfma16x16x32(const bfrag8& a, const bfrag8& b, const f32x4v& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hfrag8, a), __builtin_bit_cast(hfrag8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// per-lane epilogue constants of the 16x16 accumulator layout: bias / BatchNorm scale / shift of the lane's four columns
struct Epi16Prm { float bias[4], sc[4], sh[4]; };
__device__ __forceinline__ Epi16Prm epi16_load(const TdnnParams& p, int n0, int wn, int lane) {
    Epi16Prm e;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        const bool nv = n < p.units;
        e.bias[j] = (nv && p.bias) ? p.bias[n] : 0.0f;
        e.sc[j] = (nv && p.scale) ? p.scale[n] : 1.0f;
        e.sh[j] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }
    return e;
}

// 16-byte store of a piece of a 16-bit activation plane. The plane (1 GB per layer at 1024 utterances) is read by the NEXT
// launch only: written non-temporally it does not push the weights and the activation tiles two workgroups share out of the
// XCD's L2 (+1.2 % on the whole step; non-temporal operand LOADS cost 3-6 %).
__device__ __forceinline__ void st16(u32x4* dst, const u32x4& v) {
    if (KTF_X3_Y_NT) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}

template <int ACT, bool STATS, bool F16 = false>
__device__ __forceinline__ void ring_epilogue16(f32x4v (&acc)[8][4], const TdnnParams& p, double* __restrict__ stats,
                                                unsigned char* rsm, int b, int t0, int n0, int out_len, int wm, int wn,
                                                int wave, int lane, const Epi16Prm& prm) {
    float* et = reinterpret_cast<float*>(rsm);
    const float (&bias)[4] = prm.bias;
    const float (&sc)[4] = prm.sc;
    const float (&sh)[4] = prm.sh;
    const int rows_valid = out_len - t0;
    if (STATS) {
        // A lane holds 32 rows of each of its 4 columns. Their sum and sum of squares are taken in fp32 RELATIVE TO A PIVOT
        // (row 0 of the wave's 128-row block: a constant column -- a dead ReLU unit -- gives exactly 0 and 0, not fp32
        // cancellation noise) and only the per-lane results go to fp64: 32 x 3 fp32 operations per column instead of 32 x 3
        // fp64 ones (the fp64 form was 4.3 us per tile, a fifth of a K = 512 tile's K-loop).
        const int rv = rows_valid - wm * 128;                  // valid rows of this wave's block (may be <= 0)
        const int g4 = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v0 = acc[0][j][0] + bias[j];
            if (ACT == KTF_ACT_RELU) v0 = fmaxf(v0, 0.0f);
            else if (ACT != KTF_ACT_NONE) v0 = apply_act(v0, ACT);
            v0 = v0 * sc[j] + sh[j];
            const float pv = __shfl(v0, lane & 15, 64);          // row 0 of the block lives in the g4 == 0 lane of this column
            float s32 = 0.0f, q32 = 0.0f;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bias[j];
                    if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                    v = v * sc[j] + sh[j];
                    if (rv >= 128 || i * 16 + g4 * 4 + r < rv) {       // first term wave-uniform: full blocks carry no row predicate
                        const float u = v - pv;
                        s32 += u;
                        q32 = fmaf(u, u, q32);
                        ++cnt;
                    }
                }
            }
            const double pd = (double)pv, sd = (double)s32, nd = (double)cnt;
            double s = sd + nd * pd;
            double q = (double)q32 + 2.0 * pd * sd + nd * pd * pd;
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);      // the four 16-lane groups hold the same column
            s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
            const int n = n0 + wn * 64 + j * 16 + (lane & 15);
            if (lane < 16 && n < p.units) stats_out(stats, p, b, (t0 >> 7) + wm, n, s, q);
        }
        return;
    }
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    const int nl = lane * 4;
    const int n = n0 + nl;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {          // rows wm*128 + pass*32 .. +31 of both wave rows -> 64 staged rows
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
            const int i = pass * 2 + ih;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = wn * 64 + j * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int srow = wm * 32 + ih * 16 + (lane >> 4) * 4 + r;
                    float v = acc[i][j][r] + bias[j];
                    if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                    v = v * sc[j] + sh[j];
                    et[srow * R_EPI_PITCH + col] = v;
                }
            }
        }
        __syncthreads();
        if (p.y_dtype != KTF_F32 && p.ychunk) {
            // chunk-major 16-bit output: an instruction stores 16 rows of ONE 32-column chunk = 1 KiB of consecutive bytes.
            // Columns beyond `units` inside the last chunk are stored too: they are exact zeros (zero weight rows, no bias),
            // which is what the consumer's pad columns must hold.
            const int piece = lane & 3, rr = lane >> 2;
            const int64_t nchy = p.ldy >> 5;
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int item = sp * 8 + wave;                       // (chunk of the tile, group of 16 staged rows)
                const int cidx = item & 7, srow = (item >> 3) * 16 + rr;
                const int m = (srow >> 5) * 128 + pass * 32 + (srow & 31);
                const int n8 = n0 + cidx * 32 + piece * 8;
                if (m < rows_valid && n8 < p.ldy) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + cidx * 32 + piece * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + cidx * 32 + piece * 8 + 4);
                    const int64_t off = (((int64_t)b * nchy + (n8 >> 5)) * p.Tout + (t0 + m)) * 32 + (n8 & 31);
                    const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned short hh[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) hh[e] = f2x16<F16>(F16 ? fminf(fmaxf(vv[e], -65504.0f), 65504.0f) : vv[e]);   // half planes saturate instead of overflowing to inf
                    u32x4 pk;
                    pk.x = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
                    pk.y = (unsigned)hh[2] | ((unsigned)hh[3] << 16);
                    pk.z = (unsigned)hh[4] | ((unsigned)hh[5] << 16);
                    pk.w = (unsigned)hh[6] | ((unsigned)hh[7] << 16);
                    st16(reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.y) + off), pk);
                    if (p.y_lo) {
                        u32x4 pl;
                        pl.x = (unsigned)f2bf(vv[0] - bf2f(hh[0])) | ((unsigned)f2bf(vv[1] - bf2f(hh[1])) << 16);
                        pl.y = (unsigned)f2bf(vv[2] - bf2f(hh[2])) | ((unsigned)f2bf(vv[3] - bf2f(hh[3])) << 16);
                        pl.z = (unsigned)f2bf(vv[4] - bf2f(hh[4])) | ((unsigned)f2bf(vv[5] - bf2f(hh[5])) << 16);
                        pl.w = (unsigned)f2bf(vv[6] - bf2f(hh[6])) | ((unsigned)f2bf(vv[7] - bf2f(hh[7])) << 16);
                        st16(reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.y_lo) + off), pl);
                    }
                }
            }
        } else if (p.y_dtype != KTF_F32) {
            // bf16 output: 16-byte stores (8 columns per lane, two staged rows per wave instruction)
            const int n8 = n0 + (lane & 31) * 8;
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int srow = sp * 16 + wave * 2 + (lane >> 5);
                const int m = (srow >> 5) * 128 + pass * 32 + (srow & 31);
                if (m < rows_valid) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + (lane & 31) * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + (lane & 31) * 8 + 4);
                    const int64_t off = (out_row0 + m) * p.ldy + n8;
                    unsigned short* yp = reinterpret_cast<unsigned short*>(p.y) + off;
                    const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned short hh[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) hh[e] = f2x16<F16>(F16 ? fminf(fmaxf(vv[e], -65504.0f), 65504.0f) : vv[e]);   // half planes saturate instead of overflowing to inf
                    if (n8 + 8 <= p.units) {
                        u32x4 pk;
                        pk.x = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
                        pk.y = (unsigned)hh[2] | ((unsigned)hh[3] << 16);
                        pk.z = (unsigned)hh[4] | ((unsigned)hh[5] << 16);
                        pk.w = (unsigned)hh[6] | ((unsigned)hh[7] << 16);
                        st16(reinterpret_cast<u32x4*>(yp), pk);
                        if (p.y_lo) {            // split-bf16 output: the residual plane, the next layer's lo operand
                            u32x4 pl;
                            pl.x = (unsigned)f2bf(vv[0] - bf2f(hh[0])) | ((unsigned)f2bf(vv[1] - bf2f(hh[1])) << 16);
                            pl.y = (unsigned)f2bf(vv[2] - bf2f(hh[2])) | ((unsigned)f2bf(vv[3] - bf2f(hh[3])) << 16);
                            pl.z = (unsigned)f2bf(vv[4] - bf2f(hh[4])) | ((unsigned)f2bf(vv[5] - bf2f(hh[5])) << 16);
                            pl.w = (unsigned)f2bf(vv[6] - bf2f(hh[6])) | ((unsigned)f2bf(vv[7] - bf2f(hh[7])) << 16);
                            st16(reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.y_lo) + off), pl);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (n8 + e < p.units) {
                                yp[e] = hh[e];
                                if (p.y_lo) reinterpret_cast<unsigned short*>(p.y_lo)[off + e] = f2bf(vv[e] - bf2f(hh[e]));
                            }
                    }
                }
            }
        } else
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
            const int srow = sp * 8 + wave;
            const int m = (srow >> 5) * 128 + pass * 32 + (srow & 31);
            if (m < rows_valid) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(et + srow * R_EPI_PITCH + nl);
                const int64_t off = (out_row0 + m) * p.ldy + n;
                if (n + 4 <= p.units) {
                    if (p.y_dtype == KTF_F32) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + off) = v;
                    } else {
                        uint2 pk;
                        pk.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
                        pk.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + off) = pk;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.units) {
                            if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off + e] = v[e];
                            else reinterpret_cast<unsigned short*>(p.y)[off + e] = f2bf(v[e]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// Non-reducing epilogue of the 16x16x32 kernel. The MFMA operands are swapped there (W fragment as A, x fragment as B), so
// a lane's four accumulator values are four CONSECUTIVE output columns of one output row:
//   acc[i][j][r] = out[row wm*128 + i*16 + (lane&15)][col wn*64 + j*16 + (lane>>4)*4 + r]
// bias/ReLU/BatchNorm, the bf16 pack and the store therefore need no LDS staging and no barrier; the four stores of one i
// (j = 0..3) complete a 128-byte line of each of the 16 rows.
template <int ACT>
__device__ __forceinline__ void ring_epilogue16_direct(f32x4v (&acc)[8][4], const TdnnParams& p, int b, int t0, int n0,
                                                       int out_len, int wm, int wn, int lane) {
    const int c = lane & 15, g = lane >> 4;
    f32x4v bias[4], sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wn * 64 + j * 16 + g * 4 + e;
            const bool nv = n < p.units;
            bias[j][e] = (nv && p.bias) ? p.bias[n] : 0.0f;
            sc[j][e] = (nv && p.scale) ? p.scale[n] : 1.0f;
            sh[j][e] = (nv && p.shift) ? p.shift[n] : 0.0f;
        }
    }
    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = wm * 128 + i * 16 + c;
        if (m >= rows_valid) continue;
        const int64_t rowoff = (out_row0 + m) * p.ldy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + g * 4;
            f32x4v v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[i][j][e] + bias[j][e];
                if (ACT == KTF_ACT_RELU) t = fmaxf(t, 0.0f);
                else if (ACT != KTF_ACT_NONE) t = apply_act(t, ACT);
                v[e] = t * sc[j][e] + sh[j][e];
            }
            const int64_t off = rowoff + n;
            if (n + 4 <= p.units) {
                if (p.y_dtype == KTF_F32) {
                    *reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.y) + off) = v;
                } else {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                    pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + off) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < p.units) {
                        if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off + e] = v[e];
                        else reinterpret_cast<unsigned short*>(p.y)[off + e] = f2bf(v[e]);
                    }
                }
            }
        }
    }
}

// bf16-output epilogue of the 16x16x32 kernel (swapped operands, see ring_epilogue16_direct): bias/ReLU/BatchNorm and the
// bf16 pack happen in registers, each lane stages its four consecutive columns with one ds_write_b64 (row pitch 520 B: the
// 16 lanes of a store group cover all 32 banks), and after ONE barrier every wave streams 32 staged rows out with 16-byte
// stores (two 512-byte rows per wave instruction). The stores are issue-bound per instruction (T21), hence the wide form.
#define R16_PK_PITCH 520
#define R16_PRM_OFF (R_BM * R16_PK_PITCH)              // bias | scale | shift of the tile's 256 columns, behind the staging image
#define R16_LDS_BYTES (R16_PRM_OFF + 3 * R_BN * 4)      // 136,192 B

// every wave streams 32 rows of the staged 256 x 256 16-bit image out with 16-byte stores (two 512-byte rows per instruction)
__device__ __forceinline__ void r16_store_staged(const TdnnParams& p, const unsigned char* rsm, unsigned short* ybase, int b,
                                                 int t0, int n0, int out_len, int wave, int lane) {
    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    const int n8 = n0 + (lane & 31) * 8;
    const bool wide = (n8 + 8 <= p.units) && ((p.ldy & 7) == 0) && ((reinterpret_cast<uintptr_t>(ybase) & 15) == 0);
#pragma unroll 4
    for (int sp = 0; sp < 16; ++sp) {
        const int m = wave * 32 + sp * 2 + (lane >> 5);
        if (m < rows_valid) {
            const unsigned char* src = rsm + m * R16_PK_PITCH + (lane & 31) * 16;
            const uint2 lo = *reinterpret_cast<const uint2*>(src);
            const uint2 hi = *reinterpret_cast<const uint2*>(src + 8);
            unsigned short* yp = ybase + (out_row0 + m) * p.ldy + n8;
            if (wide) {
                u32x4 o;
                o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
                *reinterpret_cast<u32x4*>(yp) = o;
            } else {
                const unsigned w4[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n8 + e < p.units) yp[e] = (unsigned short)(w4[e >> 1] >> ((e & 1) * 16));
            }
        }
    }
}
template <int ACT, bool F16>
__device__ __forceinline__ void ring_epilogue16_pk(f32x4v (&acc)[8][4], const TdnnParams& p, unsigned char* rsm, int b,
                                                   int t0, int n0, int out_len, int wm, int wn, int wave, int lane) {
    const int c = lane & 15, g = lane >> 4;
    // column constants were parked in LDS when the tile started (no global loads, and no latency, at this point)
    const float* prm = reinterpret_cast<const float*>(rsm + R16_PRM_OFF);
    f32x4v bias[4], sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + g * 4;
        bias[j] = *reinterpret_cast<const f32x4v*>(prm + nl);
        sc[j] = *reinterpret_cast<const f32x4v*>(prm + R_BN + nl);
        sh[j] = *reinterpret_cast<const f32x4v*>(prm + 2 * R_BN + nl);
    }
    unsigned char* stg = rsm + (wm * 128 + c) * R16_PK_PITCH + (wn * 64 + g * 4) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4v v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = acc[i][j][e] + bias[j][e];
                if (ACT == KTF_ACT_RELU) t = fmaxf(t, 0.0f);
                else if (ACT != KTF_ACT_NONE) t = apply_act(t, ACT);
                v[e] = t * sc[j][e] + sh[j][e];
            }
            uint2 pk;
            pk.x = (unsigned)f2x16<F16>(v[0]) | ((unsigned)f2x16<F16>(v[1]) << 16);
            pk.y = (unsigned)f2x16<F16>(v[2]) | ((unsigned)f2x16<F16>(v[3]) << 16);
            *reinterpret_cast<uint2*>(stg + i * 16 * R16_PK_PITCH + j * 32) = pk;
        }
    }
    __syncthreads();
    r16_store_staged(p, rsm, reinterpret_cast<unsigned short*>(p.y), b, t0, n0, out_len, wave, lane);
}

template <int ACT, bool STATS, bool F16>
__device__ __forceinline__ void r16_tile(const TdnnParams& p, int mtiles, int ntiles, int gtiles,
                                         double* __restrict__ stats, unsigned char* rsm, const int id) {
    const int xcd = id & 7, slot = id >> 3;
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = g / mtiles, mt = g - b * mtiles;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && nt == 0 && mt == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = mt * R_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = nt * R_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    if (!STATS && tid < R_BN) {
        float* prm = reinterpret_cast<float*>(rsm + R16_PRM_OFF);
        const int n = n0 + tid;
        const bool nv = n < p.units;
        prm[tid] = (nv && p.bias) ? p.bias[n] : 0.0f;
        prm[R_BN + tid] = (nv && p.scale) ? p.scale[n] : 1.0f;
        prm[2 * R_BN + tid] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }

    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * 2;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const unsigned ldxb = (unsigned)p.ldx * 2u;

    int a_t[2];
    unsigned a_cb[2], w_ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        const int row = q >> 2;
        const unsigned chunk = (unsigned)(((q & 3) ^ ((4 - ((row >> 2) & 3)) & 3)) * 16);
        a_cb[i] = chunk;
        a_t[i] = start + (t0 + row) * p.sub;
        w_ob[i] = (unsigned)(n0 + row) * (unsigned)p.ktot * 2u + chunk;
    }

    f32x4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / R_BK;
    const int lenm1 = len - 1;
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 2;
#define S_DMA_A(i)                                                                                                     \
    {                                                                                                                  \
        int r_ = a_t[i] + is_off;                                                                                      \
        r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                                   \
        const unsigned vo_ = (unsigned)r_ * ldxb + a_cb[i] + (unsigned)is_db;                                          \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + vo_),                                                       \
            (lds_ptr_t*)(rsm + (is_ks & (R_NSTAGE - 1)) * R_STAGE_BYTES + wave * 1024 + (i) * 8192), 16, 0, KTF_AUX_A);\
    }
#define S_DMA_B(i)                                                                                                     \
    {                                                                                                                  \
        const unsigned vo_ = w_ob[i] + (unsigned)(is_ks * (R_BK * 2));                                                 \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + vo_),                                                       \
            (lds_ptr_t*)(rsm + (is_ks & (R_NSTAGE - 1)) * R_STAGE_BYTES + R_TILE_BYTES + wave * 1024 + (i) * 8192),    \
            16, 0, KTF_AUX_W);                                                                                            \
    }
#define S_ADVANCE()                                                                                                    \
    {                                                                                                                  \
        ++is_ks;                                                                                                       \
        is_db += R_BK * 2;                                                                                             \
        if (is_db == dpad_b) {                                                                                         \
            is_db = 0;                                                                                                 \
            ++is_c;                                                                                                    \
            is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                                \
        }                                                                                                              \
    }
    for (int s_ = 0; s_ < 3 && s_ < nk; ++s_) {
        S_DMA_A(0) S_DMA_B(0) S_DMA_A(1) S_DMA_B(1)
        S_ADVANCE()
    }
    // fragment addressing: lane (r = lane&15, c = lane>>4) reads row R, chunk c ^ f(R); all tile rows keep (R>>2)&3 of r
    const int fr = (4 - (((lane & 15) >> 2) & 3)) & 3;
    const int coff = (((lane >> 4) ^ fr) << 4);
    const int a_row_off = (wm * 128 + (lane & 15)) * 64 + coff;
    const int b_row_off = (wn * 64 + (lane & 15)) * 64 + coff;
    bfrag8 a[8], bq[4];
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool refill = is_ks < nk;
        const unsigned char* sa = rsm + (ks & (R_NSTAGE - 1)) * R_STAGE_BYTES;
        const unsigned char* sb = sa + R_TILE_BYTES;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i * 16 * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const bfrag8*>(sb + b_row_off + j * 16 * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        // first half: rows 0-63 of the wave's block; the second half's A fragments are fetched behind the first MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = STATS ? mfma16x16x32<F16>(a[i], bq[j], acc[i][j]) : mfma16x16x32<F16>(bq[j], a[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {
#pragma unroll
                for (int i2 = 4; i2 < 8; ++i2) a[i2] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i2 * 16 * 64);
            }
            if (refill) {
                if (i == 1) S_DMA_A(0)
                if (i == 3) S_DMA_B(0)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 4; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = STATS ? mfma16x16x32<F16>(a[i], bq[j], acc[i][j]) : mfma16x16x32<F16>(bq[j], a[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 4 && ks + 1 < nk) {
                // next stage (certified by this K-step's barrier): first-half A fragments; a[0..3] are no longer needed
                const unsigned char* san = rsm + ((ks + 1) & (R_NSTAGE - 1)) * R_STAGE_BYTES;
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) a[i2] = *reinterpret_cast<const bfrag8*>(san + a_row_off + i2 * 16 * 64);
            }
            if (refill) {
                if (i == 5) S_DMA_A(1)
                if (i == 7) S_DMA_B(1)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ks + 1 < nk) {
            // B fragments of the next stage: all MFMAs of this stage have been issued
            const unsigned char* sbn = rsm + ((ks + 1) & (R_NSTAGE - 1)) * R_STAGE_BYTES + R_TILE_BYTES;
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const bfrag8*>(sbn + b_row_off + j * 16 * 64);
        }
        if (is_ks < nk) S_ADVANCE()
    }
#undef S_DMA_A
#undef S_DMA_B
#undef S_ADVANCE
    if (STATS) {
        ring_epilogue16<ACT, STATS>(acc, p, stats, rsm, b, t0, n0, out_len, wm, wn, wave, lane, epi16_load(p, n0, wn, lane));
    } else if (p.y_dtype == KTF_F32) {
        ring_epilogue16_direct<ACT>(acc, p, b, t0, n0, out_len, wm, wn, lane);
    } else {
        __syncthreads();          // every wave's fragment reads are done before the ring is reused as staging
        ring_epilogue16_pk<ACT, F16>(acc, p, rsm, b, t0, n0, out_len, wm, wn, wave, lane);
    }
}

template <int ACT, bool STATS, bool F16>
__global__ __launch_bounds__(512) void tdnn_bf16r16_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
                                                           double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    r16_tile<ACT, STATS, F16>(p, mtiles, ntiles, gtiles, stats, rsm, blockIdx.x);
}

// ------------------------------------------------------------------------------------ BF16, 128x256 tile, 2 workgroups/CU
// The 256x256 kernel keeps one workgroup per CU, so its fixed per-tile phases (address setup, first-stage latency,
// epilogue: ~11 us against 16 us of K-loop at K = 512) leave the MFMA pipe idle. Here a workgroup is FOUR waves on a
// 128 x 256 tile (the same 128 x 64 block of 16x16x32 MFMAs per wave) with a 3-stage ring of 24 KiB stages: 76,800 B of
// LDS and <= 256 VGPRs let TWO workgroups share a CU, out of phase, so one's epilogue / prologue overlaps the other's
// K-loop and the two waves of a SIMD are no longer barrier-locked to each other.
//  * bias is preloaded into the accumulators, BatchNorm scale/shift sit in LDS (no global loads in the epilogue);
//  * non-reducing epilogue: operands swapped (W fragment as A) so a lane owns 4 consecutive columns -> packed bf16
//    ds_write_b64 staging, one barrier, 16-byte global stores (store issue is per instruction, T21);
//  * reducing (fused StatsPooling) epilogue: natural operand order, fp64 column sums, fp64 atomics.
#define H_BM 128
#define H_BN 256
#define H_NSTAGE 3
#define H_A_BYTES (H_BM * R_BK * 2)                  // 8 KiB
#define H_B_BYTES (H_BN * R_BK * 2)                  // 16 KiB
#define H_STAGE_BYTES (H_A_BYTES + H_B_BYTES)        // 24 KiB
#define H_RING_BYTES (H_NSTAGE * H_STAGE_BYTES)      // 72 KiB (bf16 staging of the tile: 128 x 520 B = 66,560 B)
#define H_LDS_BYTES (H_RING_BYTES + 3 * H_BN * 4)    // + bias | scale | shift of the tile's columns = 76,800 B
#define H_PK_PITCH 520

#ifndef KTF_H_ABL
#define KTF_H_ABL 0   // timing-only ablations of the K-loop (tools/tile_probe.py; results are garbage): 1 no refill DMA,
                    // 2 also no LDS reads, 3 no MFMA, 5 DMA issued but never waited for, 6 DMA from one small hot region (7: W only, 8: A only), 9 real addresses but 128-byte pieces (what a 64-deep K-step would fetch), 10 activation DMAs only in N-tile 0
#endif
#ifdef KTF_TILE_PROBE
#define H_PROBE(k) if (dbgp && threadIdx.x == 0) dbgp[k] = wall_clock64();
#define H_PROBE_HW()                                                                  \
    if (dbgp && threadIdx.x == 0) {                                                   \
        unsigned hw_, xcc_;                                                           \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));             \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));           \
        dbgp[5] = hw_; dbgp[6] = xcc_; dbgp[7] = nk;                                  \
    }
#else
#define H_PROBE_HW()
#define H_PROBE(k)
#endif

template <int ACT, bool STATS, bool F16>
__global__ __launch_bounds__(256, 2) void tdnn_bf16h_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
                                                            double* __restrict__ stats, long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const int id = blockIdx.x;
#ifdef KTF_TILE_PROBE
    long long* dbgp = dbg ? dbg + (int64_t)id * 16 : nullptr;
#endif
    H_PROBE(0)
    const int xcd = id & 7, slot = id >> 3;
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = g / mtiles, mt = g - b * mtiles;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && nt == 0 && mt == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = mt * H_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = nt * H_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g4 = lane >> 4;

    float* prm = reinterpret_cast<float*>(rsm + H_RING_BYTES);
    {
        const int n = n0 + tid;
        const bool nv = n < p.units;
        prm[tid] = (nv && p.bias) ? p.bias[n] : 0.0f;
        prm[H_BN + tid] = (nv && p.scale) ? p.scale[n] : 1.0f;
        prm[2 * H_BN + tid] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }

    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * 2;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const unsigned ldxb = (unsigned)p.ldx * 2u;

    // LDS-DMA chunk q = i*256 + tid -> tile row q/4, LDS position q%4, global chunk (q%4) ^ f(row) (f as in the r16 kernel)
    int a_t[2];
    unsigned a_cb[2], w_ob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid;
        const int row = q >> 2;
        const unsigned chunk = (unsigned)(((q & 3) ^ ((4 - ((row >> 2) & 3)) & 3)) * 16);
        if (i < 2) {
            a_cb[i] = chunk;
            a_t[i] = start + (t0 + row) * p.sub;
        }
        w_ob[i] = (unsigned)(n0 + row) * (unsigned)p.ktot * 2u + chunk;
    }

    const int nk = p.ktot / R_BK;
    const int lenm1 = len - 1;
    int is_ks = 0, is_slot = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 2;
#define H_DMA_A(i)                                                                                                     \
    {                                                                                                                  \
        int r_ = a_t[i] + is_off;                                                                                      \
        r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                                   \
        unsigned vo_ = (KTF_H_ABL == 6 || KTF_H_ABL == 8) ? a_cb[i] + (unsigned)(tid >> 2) * 64u                                   \
                                              : (unsigned)r_ * ldxb + a_cb[i] + (unsigned)is_db;                       \
        if (KTF_H_ABL == 9) {                                                                                          \
            const int q_ = (i) * 256 + tid;                                                                            \
            int r9_ = start + t0 + (q_ >> 3) + ((is_ks & 1) ? 64 : 0) + is_off;                                        \
            r9_ = r9_ < 0 ? 0 : (r9_ > lenm1 ? lenm1 : r9_);                                                           \
            vo_ = (unsigned)r9_ * ldxb + ((unsigned)is_db & ~127u) + (unsigned)(q_ & 7) * 16u;                         \
        }                                                                                                              \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + vo_),                                                       \
            (lds_ptr_t*)(rsm + is_slot * H_STAGE_BYTES + wn * 1024 + (i) * 4096), 16, 0, KTF_AUX_A);                   \
    }
#define H_DMA_B(i)                                                                                                     \
    {                                                                                                                  \
        unsigned vo_ = (KTF_H_ABL == 6 || KTF_H_ABL == 7) ? (unsigned)((i) * 256 + tid) * 16u                                      \
                                              : w_ob[i] + (unsigned)(is_ks * (R_BK * 2));                              \
        if (KTF_H_ABL == 9) {                                                                                          \
            const int q_ = (i) * 256 + tid;                                                                            \
            vo_ = (unsigned)(n0 + (q_ >> 3) + ((is_ks & 1) ? 128 : 0)) * (unsigned)p.ktot * 2u +                       \
                  ((unsigned)(is_ks * (R_BK * 2)) & ~127u) + (unsigned)(q_ & 7) * 16u;                                 \
        }                                                                                                              \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + vo_),                                                       \
            (lds_ptr_t*)(rsm + is_slot * H_STAGE_BYTES + H_A_BYTES + wn * 1024 + (i) * 4096), 16, 0, KTF_AUX_W);       \
    }
#define H_ADVANCE()                                                                                                    \
    {                                                                                                                  \
        ++is_ks;                                                                                                       \
        is_slot = (is_slot == H_NSTAGE - 1) ? 0 : is_slot + 1;                                                         \
        is_db += R_BK * 2;                                                                                             \
        if (is_db == dpad_b) {                                                                                         \
            is_db = 0;                                                                                                 \
            ++is_c;                                                                                                    \
            is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                                \
        }                                                                                                              \
    }
    for (int s_ = 0; s_ < 2 && s_ < nk; ++s_) {
        H_DMA_A(0) H_DMA_A(1) H_DMA_B(0) H_DMA_B(1) H_DMA_B(2) H_DMA_B(3)
        H_ADVANCE()
    }
    H_PROBE(1)
    __syncthreads();                                  // prm[] visible

    f32x4v acc[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4v bj;
        if (STATS) {
            const float bv = prm[wn * 64 + j * 16 + c];
            bj[0] = bv; bj[1] = bv; bj[2] = bv; bj[3] = bv;
        } else {
            bj = *reinterpret_cast<const f32x4v*>(prm + wn * 64 + j * 16 + g4 * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = bj;
    }

    const int fr = (4 - ((c >> 2) & 3)) & 3;
    const int coff = ((g4 ^ fr) << 4);
    const int a_row_off = c * 64 + coff;
    const int b_row_off = (wn * 64 + c) * 64 + coff;
    bfrag8 a[8], bq[4];
    int cs = 0;
    for (int ks = 0; ks < nk; ++ks) {
        if (KTF_H_ABL != 5) {
            if (KTF_H_ABL == 10 && nt != 0 && ks + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef KTF_TILE_PROBE
        if (ks == 0) { H_PROBE(2) if (dbgp && threadIdx.x == 0) dbgp[12] = clock64(); }
#endif
        const bool refill = is_ks < nk && (KTF_H_ABL == 0 || KTF_H_ABL >= 3);
        const unsigned char* sa = rsm + cs * H_STAGE_BYTES;
        const unsigned char* sb = sa + H_A_BYTES;
        cs = (cs == H_NSTAGE - 1) ? 0 : cs + 1;
        if (KTF_H_ABL != 2 || ks == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i * 16 * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const bfrag8*>(sb + b_row_off + j * 16 * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (KTF_H_ABL != 3)
                acc[i][j] = STATS ? mfma16x16x32<F16>(a[i], bq[j], acc[i][j]) : mfma16x16x32<F16>(bq[j], a[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0 && (KTF_H_ABL != 2 || ks == 0)) {
#pragma unroll
                for (int i2 = 4; i2 < 8; ++i2) a[i2] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i2 * 16 * 64);
            }
            if (refill) {
                if (i == 1 && (KTF_H_ABL != 10 || nt == 0)) H_DMA_A(0)
                if (i == 2 && (KTF_H_ABL != 10 || nt == 0)) H_DMA_A(1)
                if (i == 3) H_DMA_B(0)
                if (i == 4) H_DMA_B(1)
                if (i == 5) H_DMA_B(2)
                if (i == 6) H_DMA_B(3)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (refill) H_ADVANCE()
    }
#undef H_DMA_A
#undef H_DMA_B
#undef H_ADVANCE
    if (KTF_H_ABL == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    H_PROBE(3)
#ifdef KTF_TILE_PROBE
    if (dbgp && threadIdx.x == 0) dbgp[13] = clock64();
#endif
    const int rows_valid = out_len - t0;
    if (STATS) {
        // acc[i][j][r] = out[row i*16 + g4*4 + r][col wn*64 + j*16 + c]
        // Column sums with a PIVOT: every lane accumulates sum(v - p) and sum((v - p)^2) in fp32, where p is the column's
        // value in the tile's first row (the same for the four lanes that share a column), and converts to the absolute
        // sums in fp64 once per tile: sum v = s + n p, sum v^2 = q + 2 p s + n p^2. A constant channel (dead ReLU, zero
        // weight row) gives v - p == 0 exactly, hence var == 0 exactly as with fp64 accumulation of v, v^2 -- at 5 fp32
        // operations per element instead of 2 fp32 + 3 fp64.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + c;
            const float scj = prm[H_BN + nl], shj = prm[2 * H_BN + nl];
            float v0 = acc[0][j][0];
            if (ACT == KTF_ACT_RELU) v0 = fmaxf(v0, 0.0f);
            else if (ACT != KTF_ACT_NONE) v0 = apply_act(v0, ACT);
            v0 = v0 * scj + shj;
            const float pv = __shfl(v0, c, 64);       // row 0 of the tile lives in the g4 == 0 lane of this column
            float s32 = 0.0f, q32 = 0.0f;
            int cnt = 0;
            if (rows_valid >= H_BM) {                 // wave-uniform: full tiles carry no row predicate
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r];
                        if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                        else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                        v = v * scj + shj;
                        const float u = v - pv;
                        s32 += u;
                        q32 = fmaf(u, u, q32);
                    }
                }
                cnt = 32;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = i * 16 + g4 * 4 + r;
                        float v = acc[i][j][r];
                        if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                        else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                        v = v * scj + shj;
                        if (m < rows_valid) {
                            const float u = v - pv;
                            s32 += u;
                            q32 = fmaf(u, u, q32);
                            ++cnt;
                        }
                    }
                }
            }
            const double pd = (double)pv, sd = (double)s32, nd = (double)cnt;
            double sm = sd + nd * pd;
            double sq = (double)q32 + 2.0 * pd * sd + nd * pd * pd;
            sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
            sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
            const int n = n0 + nl;
            if (lane < 16 && n < p.units) stats_out(stats, p, b, t0 >> 7, n, sm, sq);
        }
        H_PROBE(4)
        H_PROBE_HW()
        return;
    }
    // acc[i][j][e] = out[row i*16 + c][col wn*64 + j*16 + g4*4 + e]
    f32x4v sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = *reinterpret_cast<const f32x4v*>(prm + H_BN + wn * 64 + j * 16 + g4 * 4);
        sh[j] = *reinterpret_cast<const f32x4v*>(prm + 2 * H_BN + wn * 64 + j * 16 + g4 * 4);
    }
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    if (p.y_dtype == KTF_F32) {
        float* ybase = reinterpret_cast<float*>(p.y);
        const bool vec_ok = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = i * 16 + c;
            if (m >= rows_valid) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4v v = acc[i][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ACT == KTF_ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
                    else if (ACT != KTF_ACT_NONE) v[e] = apply_act(v[e], ACT);
                }
                v = v * sc[j] + sh[j];
                const int n = n0 + wn * 64 + j * 16 + g4 * 4;
                float* yp = ybase + (out_row0 + m) * p.ldy + n;
                if (vec_ok && n + 4 <= p.units) {
                    *reinterpret_cast<f32x4v*>(yp) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.units) yp[e] = v[e];
                }
            }
        }
        return;
    }
    __syncthreads();                                  // all fragment reads done: the ring becomes the staging buffer
    H_PROBE(8)
    {
        unsigned char* stg = rsm + c * H_PK_PITCH + (wn * 64 + g4 * 4) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4v v = acc[i][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ACT == KTF_ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
                    else if (ACT != KTF_ACT_NONE) v[e] = apply_act(v[e], ACT);
                }
                v = v * sc[j] + sh[j];
                uint2 pk;
                pk.x = (unsigned)f2x16<F16>(v[0]) | ((unsigned)f2x16<F16>(v[1]) << 16);
                pk.y = (unsigned)f2x16<F16>(v[2]) | ((unsigned)f2x16<F16>(v[3]) << 16);
                *reinterpret_cast<uint2*>(stg + i * 16 * H_PK_PITCH + j * 32) = pk;
            }
        }
    }
    H_PROBE(9)
    __syncthreads();
    H_PROBE(10)
    {
        const int n8 = n0 + (lane & 31) * 8;
        const bool wide = (n8 + 8 <= p.units) && ((p.ldy & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
        unsigned short* ybase = reinterpret_cast<unsigned short*>(p.y);
#pragma unroll 4
        for (int sp = 0; sp < 16; ++sp) {
            const int m = wn * 32 + sp * 2 + (lane >> 5);
            if (m < rows_valid) {
                const unsigned char* src = rsm + m * H_PK_PITCH + (lane & 31) * 16;
                const uint2 lo = *reinterpret_cast<const uint2*>(src);
                const uint2 hi = *reinterpret_cast<const uint2*>(src + 8);
                unsigned short* yp = ybase + (out_row0 + m) * p.ldy + n8;
                if (wide) {
                    u32x4 o;
                    o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
                    *reinterpret_cast<u32x4*>(yp) = o;
                } else {
                    const unsigned w4[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n8 + e < p.units) yp[e] = (unsigned short)(w4[e >> 1] >> ((e & 1) * 16));
                }
            }
        }
    }
    H_PROBE(4)
    H_PROBE_HW()
}

// ------------------------------------------------------------------------------------ BF16X3, 256x256 tile
// Split-bf16 on the 256x256 structure: fp32 activations are staged RAW (256 rows x 32 k x 4 B = 128-byte rows, chunk
// permutation c ^ ((row>>1)&7)) and split into bf16 hi/lo parts in registers when the fragments are read; the weights are
// pre-split on the host into two bf16 planes. acc += hi*hi + lo*hi + hi*lo: 48 MFMAs per wave per K-step against 8 DMA
// instructions, so a plain double buffer (2 x 64 KiB) with one stage in flight covers the DMA latency.
#define X_STAGE_BYTES (32768 + 2 * R_TILE_BYTES)    // A fp32 + W hi + W lo = 64 KiB
#define X_LDS_BYTES (2 * X_STAGE_BYTES)             // 128 KiB (epilogue staging 66,560 B fits)

__device__ __forceinline__ void split_bf16x8(const f32x4& v0, const f32x4& v1, bfrag8& hi, bfrag8& lo) {
    union { bfrag8 f; unsigned u[4]; } H, Lw;
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned short h0 = f2bf(x[2 * e]), h1 = f2bf(x[2 * e + 1]);
        H.u[e] = (unsigned)h0 | ((unsigned)h1 << 16);
        const unsigned short l0 = f2bf(x[2 * e] - bf2f(h0)), l1 = f2bf(x[2 * e + 1] - bf2f(h1));
        Lw.u[e] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    hi = H.f;
    lo = Lw.f;
}

// SPLIT: the activations arrive as two bf16 planes (hi, lo) written by the producing layer's epilogue (or by
// ktf_split_bf16): the A tile is then two 16 KiB bf16 images in the W layout and the K-loop carries no conversion at
// all -- the in-register split costs ~190 VALU instructions per wave per K-step, four times redundantly per A tile
// (measured: the K-loop is 28 % shorter without it).
template <int ACT, bool STATS, bool SPLIT>
__global__ __launch_bounds__(512) void tdnn_x3r_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
                                                       double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = g / mtiles, mt = g - b * mtiles;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && nt == 0 && mt == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = mt * R_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = nt * R_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    constexpr int XB = SPLIT ? 2 : 4;                  // bytes per activation element
    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * XB;
    const char* xl = SPLIT ? reinterpret_cast<const char*>(p.x_lo) + ((int64_t)b * p.T * p.ldx) * XB : nullptr;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const char* wl = reinterpret_cast<const char*>(p.w_lo);
    const unsigned ldxb = (unsigned)p.ldx * XB;

    // A staging: chunk q = i*512 + tid (i < 4) -> row q/8, LDS position q%8, global chunk (q%8) ^ ((row>>1)&7)
    int a_t[4];
    unsigned a_cb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 512 + tid;
        if (SPLIT) {
            // two bf16 planes, each 256 rows x 4 chunks (the W layout): i = 0,1 -> hi plane rows 0-127 / 128-255, i = 2,3 -> lo
            const int row = (q & 1023) >> 2;
            a_cb[i] = (unsigned)(((q & 3) ^ ((row >> 2) & 3)) * 16);
            a_t[i] = start + (t0 + row) * p.sub;
        } else {
            const int row = q >> 3;
            a_cb[i] = (unsigned)(((q & 7) ^ ((row >> 1) & 7)) * 16);
            a_t[i] = start + (t0 + row) * p.sub;
        }
    }
    // W staging (both planes): chunk q = i*512 + tid (i < 2) -> row q/4, position q%4, global chunk (q%4) ^ ((row>>2)&3)
    unsigned w_ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        const int row = q >> 2;
        w_ob[i] = (unsigned)(n0 + row) * (unsigned)p.ktot * 2u + (unsigned)(((q & 3) ^ ((row >> 2) & 3)) * 16);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / R_BK;
    const int lenm1 = len - 1;
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * XB;
#define X_STAGE()                                                                                                      \
    {                                                                                                                  \
        unsigned char* st_ = rsm + (is_ks & 1) * X_STAGE_BYTES + wave * 1024;                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
            int r_ = a_t[i] + is_off;                                                                                  \
            r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                               \
            const unsigned vo_ = (unsigned)r_ * ldxb + a_cb[i] + (unsigned)is_db;                                      \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(((SPLIT && i >= 2) ? xl : xb) + vo_), (lds_ptr_t*)(st_ + i * 8192), 16, 0, 0); \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            const unsigned vo_ = w_ob[i] + (unsigned)(is_ks * (R_BK * 2));                                             \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + vo_), (lds_ptr_t*)(st_ + 32768 + i * 8192), 16, 0, 0);  \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wl + vo_), (lds_ptr_t*)(st_ + 32768 + R_TILE_BYTES + i * 8192), 16, 0, 0); \
        }                                                                                                              \
        ++is_ks;                                                                                                       \
        is_db += R_BK * XB;                                                                                            \
        if (is_db == dpad_b) {                                                                                         \
            is_db = 0;                                                                                                 \
            ++is_c;                                                                                                    \
            is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                                \
        }                                                                                                              \
    }

    X_STAGE()
    const int rswa = ((lane & 31) >> 1) & 7;               // A: 128-B rows
    const int rswb = ((lane & 31) >> 2) & 3;               // W: 64-B rows
    const int a_row_off = (wm * 128 + (lane & 31)) * (SPLIT ? 64 : 128);
    const int b_row_off = (wn * 64 + (lane & 31)) * 64;
    const int hsel = lane >> 5;
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stage ks landed (nothing else is in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (is_ks < nk) X_STAGE()                            // stage ks+1 -> the buffer every wave finished reading (stage ks-1)
        const unsigned char* sa = rsm + (ks & 1) * X_STAGE_BYTES;
        const unsigned char* sh = sa + 32768;
        const unsigned char* sl = sh + R_TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bfrag8 ah[4], al[4], bh[2], bl[2];
            const int cb = ((kk * 2 + hsel) ^ rswb) << 4;
            if (SPLIT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i * 32 * 64 + cb);
                    al[i] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off + i * 32 * 64 + cb);
                }
            } else {
                const int ca = kk * 4 + 2 * hsel;            // first of the two 16-B chunks holding k = 16kk + 8h .. +7
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned char* row = sa + a_row_off + i * 32 * 128;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(row + ((ca ^ rswa) << 4));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + (((ca + 1) ^ rswa) << 4));
                    split_bf16x8(v0, v1, ah[i], al[i]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const bfrag8*>(sh + b_row_off + j * 32 * 64 + cb);
                bl[j] = *reinterpret_cast<const bfrag8*>(sl + b_row_off + j * 32 * 64 + cb);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                }
        }
    }
#undef X_STAGE
    __syncthreads();
    ring_epilogue<ACT, STATS>(acc, p, stats, rsm, b, t0, n0, out_len, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------ BF16X3 on 16x16x32, split planes
// Split-bf16 with hi/lo activation planes and fused pooling on the 16x16x32 MFMA (the chip holds a higher clock on this
// shape than on 32x32x16: the K-loop is 8-10 % shorter): 256x256 tile, 8 waves of 128x64, a stage = A hi | A lo | W hi | W lo
// (4 x 16 KiB, the chunk permutation of the 16x16x32 bf16 kernel), double buffered -- 96 MFMAs per wave per K-step cover one
// stage of DMA latency. Only the reducing (fused StatsPooling) form exists: a 16x16-layout epilogue that writes hi/lo planes
// (two staged passes) measured 10 us per tile slower than the 32x32 kernel's, which cancels the K-loop gain at K <= 1536.
#ifndef KTF_X2_PK
#define KTF_X2_PK 0           // 1: packed single-barrier epilogue (PK) for the 2-pass form's half plane. Measured equal to the
                              // fp32-staged four-pass one (92.5 k vs 92.1 k x-vectors/s): both are bound by the CU's store issue
                              // rate (128 KB per tile at ~14 B/clk = the 5.2 us the epilogue takes), not by LDS or barriers
#endif
#ifndef KTF_X2_RING3
#define KTF_X2_RING3 0        // 1: three 48 KiB stages for the 2-pass form (two stages of DMA in flight). Measured: no gain over
                              // two (92.7 k vs 93.4 k x-vectors/s), and 90.5 k vs 92.4 k when the third stage is used to read the
                              // next stage's first fragments under the last MFMA group (so that a step opens with MFMAs instead of
                              // an LDS read burst): neither DMA latency nor LDS latency is what the K-step waits for. PMC
                              // (profiles/r2_pmc_f16x2.txt): the matrix pipes are busy 54 % of the kernel at ~2.1 GHz, the waves sit
                              // in s_waitcnt / s_barrier a third of their life; without the operand DMAs the same MFMAs run 27 %
                              // faster (timing-only ablation) -- the DMA instructions' issue time in the MFMA waves is the cost.
#endif
#define XS_STAGE_BYTES (4 * R_TILE_BYTES)                // 64 KiB
#define XS_LDS_BYTES (2 * XS_STAGE_BYTES)                // 128 KiB
#define X2_LDS_BYTES (KTF_X2_RING3 ? 9 * R_TILE_BYTES : XS_LDS_BYTES)     // 2-pass form: three 48 KiB stages = 144 KiB
// PIPE = 1: hand-scheduled K-step. The stage's operand DMAs are no longer issued in one burst behind the barrier (all eight
// waves then sit in DMA issue and LDS latency together while the matrix pipes idle) but one at a time between groups of six
// MFMAs, and the A fragments of row group g+1 are read while the MFMAs of group g run (two fragment register sets).
// DOFF: waves 4-7 (the SIMD partners of waves 0-3) start their DMA slots DOFF chunks later, so that partners do not sit in
// DMA issue at the same time.
#ifndef KTF_X3_PIPE
#define KTF_X3_PIPE 1
#endif
#ifndef KTF_X3_DOFF
#define KTF_X3_DOFF 0
#endif

// F16 / TERMS: the same kernel as the 2-pass half-precision mode (KTF_GEMM_F16X2): IEEE-half operands, activations as ONE
// half plane (no residual plane: the A lo DMAs, fragments and the lo*hi pass drop out; the stage keeps its layout), weights
// as hi + lo half planes: acc += x*w_hi + x*w_lo, i.e. exact weights and half-rounded activations.
// PK (2-pass form with one 16-bit output plane): the MFMA operands are swapped (W fragment as A), so a lane's four accumulator
// values are four CONSECUTIVE output columns; bias / ReLU / affine and the half pack happen in registers, the whole 256 x 256
// tile is staged as 16-bit pairs (one ds_write_b64 per accumulator quad, 520-byte pitch) behind ONE barrier and streamed out
// as 16-byte row pieces (ring_epilogue16_pk of the bf16 kernel) -- instead of four fp32-staged 64-row passes with two
// barriers each. Products commute and the K order is unchanged: the values are bit-identical to the unswapped form.
template <int ACT, bool STATS, int PIPE = KTF_X3_PIPE, bool F16 = false, int TERMS = 3, bool PK = false>
__global__ __launch_bounds__(512) void tdnn_x3s_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
                                                       double* __restrict__ stats) {
    static_assert(TERMS == 3 || (TERMS == 2 && PIPE >= 1) || (TERMS == 1 && PIPE == 1 && !PK), "the 2-pass and 1-pass forms exist for the scheduled K-steps only");
    static_assert(!PK || (TERMS == 2 && !STATS), "the packed epilogue writes one 16-bit plane");
#ifdef KTF_TILE_PROBE
    long long* xprobe = p.probe ? p.probe + (int64_t)blockIdx.x * 8 : nullptr;
#define XS_PROBE(k) if (xprobe && threadIdx.x == 0) xprobe[k] = wall_clock64();
#else
#define XS_PROBE(k)
#endif
    XS_PROBE(0)
    // LDS ring: 64 KiB stages (A hi | A lo | W hi | W lo), double buffered; the 2-pass form leaves the A lo plane unused. Its
    // 48 KiB of live data per stage would also fit THREE deep (KTF_X2_RING3: DMAs of stage k+2 issued during step k, counted
    // vmcnt at the barrier), which measured no faster.
    static_assert(PIPE != 2 || (TERMS == 2 && !PK), "the ping-pong K-loop exists for the 2-pass form");
    constexpr int NST = (TERMS == 1) ? KTF_X1_STAGES : (TERMS == 2 && (KTF_X2_RING3 || PIPE == 2) && !PK) ? 3 : 2;
    constexpr int STG = (TERMS == 1) ? 2 * R_TILE_BYTES : (NST == 3) ? 3 * R_TILE_BYTES : XS_STAGE_BYTES;      // one pass: A | W
    constexpr int WOFF = (TERMS == 1 || NST == 3) ? R_TILE_BYTES : 2 * R_TILE_BYTES;       // W hi plane inside a stage; W lo follows it
    int fill_slot = 0, cur_slot = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = g / mtiles, mt = g - b * mtiles;
    const int n0 = nt * R_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const char* wh = reinterpret_cast<const char*>(p.w);
    const char* wl = reinterpret_cast<const char*>(p.w_lo);
    unsigned a_cb[2], w_ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        const int row = q >> 2;
        const unsigned chunk = (unsigned)(((q & 3) ^ ((4 - ((row >> 2) & 3)) & 3)) * 16);
        a_cb[i] = chunk;
        w_ob[i] = p.wtiled ? (unsigned)nt * (unsigned)(p.ktot / R_BK) * (unsigned)R_TILE_BYTES + (unsigned)q * 16u
                           : (unsigned)(n0 + row) * (unsigned)p.ktot * 2u + chunk;
    }
    const unsigned w_step = p.wtiled ? (unsigned)R_TILE_BYTES : (unsigned)(R_BK * 2);      // bytes between consecutive K-steps of W
    // The W half of stage 0 depends on the kernel arguments only: it is in flight while the utterance length (a dependent
    // scalar load) and everything derived from it are still on their way (stamps: 1.6-2.0 us from entry to the last DMA of
    // stage 0, then 0.9 us until it lands, on tiles whose K = 512 loop takes 21 us).
    if (KTF_X3_WFIRST) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned char* st_ = rsm + wave * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wh + w_ob[i]), (lds_ptr_t*)(st_ + WOFF + i * 8192), 16, 0, 0);
            if (TERMS > 1) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wl + w_ob[i]), (lds_ptr_t*)(st_ + WOFF + R_TILE_BYTES + i * 8192), 16, 0, 0);
        }
    }
    // (Tried on top: the A half too, clamped to the buffer instead of the utterance -- valid while ctx[0] <= 0 -- and the
    // epilogue constants before everything: 0.7 % and 1.5 % slower.)
    if (KTF_X3_WFIRST) asm volatile("" ::: "memory");
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && nt == 0 && mt == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = mt * R_BM;
    if (t0 >= out_len || len <= 0) {
        if (KTF_X3_WFIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing lands in the LDS of a finished workgroup
        return;
    }
    const char* xh = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * 2;      // (re-pointed by a timing ablation)
    const char* xl = reinterpret_cast<const char*>(p.x_lo) + ((int64_t)b * p.T * p.ldx) * 2;
    const unsigned ldxb = (unsigned)p.ldx * 2u;
    int a_t[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_t[i] = start + (t0 + ((i * 512 + tid) >> 2)) * p.sub;
#if defined(KTF_X3_ABL) && (KTF_X3_ABL & 16)      // timing-only ablation: every tile reads the activations of tile 0 (hot in L2)
    xh = reinterpret_cast<const char*>(p.x);
#pragma unroll
    for (int i = 0; i < 2; ++i) a_t[i] = start + ((i * 512 + tid) >> 2) * p.sub;
#endif

    f32x4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / R_BK;
    const int lenm1 = len - 1;
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 2;
    // A-piece address = row * x_rm + is_xb + chunk: row-major planes x_rm = row pitch, is_xb = byte offset of the 32-feature chunk in
    // the row; chunk-major planes x_rm = 64, is_xb = chunk index * T * 64 (branch-free: both are wave-uniform scalars)
    const unsigned x_rm = p.xchunk ? 64u : ldxb;
    const unsigned x_cs = p.xchunk ? (unsigned)p.T * 64u : (unsigned)(R_BK * 2);
    unsigned is_xb = 0;
#define XS_STAGE()                                                                                                     \
    {                                                                                                                  \
        unsigned char* st_ = rsm + fill_slot * STG + wave * 1024;                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            int r_ = a_t[i] + is_off;                                                                                  \
            r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                               \
            const unsigned vo_ = (unsigned)r_ * x_rm + a_cb[i] + is_xb;                                                \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xh + vo_), (lds_ptr_t*)(st_ + i * 8192), 16, 0, 0);          \
            if (TERMS == 3) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xl + vo_), (lds_ptr_t*)(st_ + R_TILE_BYTES + i * 8192), 16, 0, 0); \
        }                                                                                                              \
        if (!(KTF_X3_WFIRST && is_ks == 0))                  /* stage 0's W half went out at kernel entry */           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            const unsigned vo_ = w_ob[i] + (unsigned)is_ks * w_step;                                                   \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wh + vo_), (lds_ptr_t*)(st_ + WOFF + i * 8192), 16, 0, 0);   \
            if (TERMS > 1) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wl + vo_), (lds_ptr_t*)(st_ + WOFF + R_TILE_BYTES + i * 8192), 16, 0, 0); \
        }                                                                                                              \
        fill_slot = (fill_slot + 1 == NST) ? 0 : fill_slot + 1;                                                        \
        ++is_ks;                                                                                                       \
        if (p.kinter) {                                                                                                \
            if (++is_c == p.nctx) {                                                                                    \
                is_c = 0;                                                                                              \
                is_db += R_BK * 2;                                                                                     \
                is_xb += x_cs;                                                                                         \
            }                                                                                                          \
            is_off = p.ctx[is_c];                                                                                      \
        } else {                                                                                                       \
            is_db += R_BK * 2;                                                                                         \
            is_xb += x_cs;                                                                                             \
            if (is_db == dpad_b) {                                                                                     \
                is_db = 0;                                                                                             \
                is_xb = 0;                                                                                             \
                ++is_c;                                                                                                \
                is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                            \
            }                                                                                                          \
        }                                                                                                              \
    }
    XS_STAGE()
    if (NST >= 3 && PIPE == 1 && nk > 1) XS_STAGE()
    if (NST >= 4 && PIPE == 1 && nk > 2) XS_STAGE()
    Epi16Prm eprm;
    if constexpr (PK) {
        if (tid < R_BN) {                                    // column constants parked behind the staging image (read in the epilogue)
            float* prm = reinterpret_cast<float*>(rsm + R16_PRM_OFF);
            const int n = n0 + tid;
            const bool nv = n < p.units;
            prm[tid] = (nv && p.bias) ? p.bias[n] : 0.0f;
            prm[R_BN + tid] = (nv && p.scale) ? p.scale[n] : 1.0f;
            prm[2 * R_BN + tid] = (nv && p.shift) ? p.shift[n] : 0.0f;
        }
    } else {
        eprm = epi16_load(p, n0, wn, lane);                  // issued here: the ~1 us of global-load latency hides under the K-loop
    }
    XS_PROBE(1)
    const int fr = (4 - (((lane & 15) >> 2) & 3)) & 3;
    const int coff = (((lane >> 4) ^ fr) << 4);
    const int a_row_off = (wm * 128 + (lane & 15)) * 64 + coff;
    const int b_row_off = (wn * 64 + (lane & 15)) * 64 + coff;
    if constexpr (PIPE == 2) {
        // Ping-pong K-loop. The two waves of a SIMD (wave w and w + 4: row halves 0 and 1 of the tile) alternate roles between
        // barriers: while one issues its 64 MFMAs of a K-step from registers, the other issues its share of the operand
        // DMAs (three stages ahead of the reads, 48 KiB stages, three-deep ring) and reads its fragments of the next stage
        // into the registers its own MFMAs just released. The matrix pipe of a SIMD is fully paced by one wave's MFMA stream
        // (64 x 16 cycles); everything that stalled a wave in the in-phase loop -- DMA issue into a busy texture addresser,
        // LDS read latency, the barrier -- now stalls the wave that is NOT feeding the pipe.
        //   group 0 (waves 0-3), step k:  MFMA(k)               | wait, barrier k |  issue(k + 3), read(k + 1)
        //   group 1 (waves 4-7), step k:  issue(k + 2), read(k) | wait, barrier k |  MFMA(k)
        // Stage k + 1 is complete at barrier k (every wave waits for its own share: all but its youngest six DMAs); the slot
        // a group refills was last read before the previous barrier (group 1) or before this one (group 0).
        const int grp = wave >> 2;                            // wave-uniform
        if (nk > 1) XS_STAGE()                                // stage 1 (group 0 issues its share of stage 2 in its first slot)
        bfrag8 af[8], bh[4], bl[4];
#if defined(KTF_X3_ABL) && (KTF_X3_ABL & 4)       // timing-only ablation: fragments are read in the first step only
#define PP_ABL4 1
#else
#define PP_ABL4 0
#endif
#if defined(KTF_X3_ABL)
#define PP_DMA_ON (!(KTF_X3_ABL & 2))
#define PP_MFMA_ON (!(KTF_X3_ABL & 8))
#define PP_SLEEP() { if (KTF_X3_ABL & 32) { _Pragma("unroll") for (int z = 0; z < 16; ++z) __builtin_amdgcn_s_sleep(1); } }   /* 32: the MFMA segment idles for about as long instead */
#else
#define PP_SLEEP() {}
#define PP_DMA_ON 1
#define PP_MFMA_ON 1
#endif
#define PP_VM(v_) (((v_) & 15) | (((v_) >> 4) << 14) | 0x0f70)
        // own DMAs of stage `need_` have landed: all but the 6 (is_ks - 1 - need_) youngest are complete
#define PP_WAIT(need_)                                                                                                 \
    {                                                                                                                  \
        const int n__ = is_ks - 1 - (need_);                                                                           \
        if (n__ >= 2) __builtin_amdgcn_s_waitcnt(PP_VM(12));                                                           \
        else if (n__ == 1) __builtin_amdgcn_s_waitcnt(PP_VM(6));                                                       \
        else __builtin_amdgcn_s_waitcnt(PP_VM(0));                                                                     \
    }
        PP_WAIT(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        XS_PROBE(2)
        // reads are unconditional (a clamped stage index in the last step) and each group has a straight-line loop of its own:
        // a fragment register defined on one side of a branch only is a phi, and the compiler then keeps TWO fragment sets
        // (read into one, copy to the other: 64 more VGPRs, spills inside the MFMA stream)
#define PP_READ(j_)                                                                                                    \
    {                                                                                                                  \
        const unsigned char* sa_ = rsm + (PP_ABL4 ? 0 : ((j_) % 3) * STG);                                             \
        const unsigned char* sw_ = sa_ + WOFF;                                                                         \
        if (!PP_ABL4 || (j_) == 0)                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bh[j] = *reinterpret_cast<const bfrag8*>(sw_ + b_row_off + j * 16 * 64); \
        if (!PP_ABL4 || (j_) == 0)                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bfrag8*>(sa_ + a_row_off + i * 16 * 64); \
        if (!PP_ABL4 || (j_) == 0)                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const bfrag8*>(sw_ + R_TILE_BYTES + b_row_off + j * 16 * 64); \
        __builtin_amdgcn_s_waitcnt(0xc07f);     /* lgkmcnt(0), visible to the compiler's counter model: complete before the barrier that releases the slot's refill */ \
    }
#define PP_MFMA()                                                                                                      \
    {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x16x32<F16>(af[i], bh[j], acc[i][j]);      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x16x32<F16>(af[i], bl[j], acc[i][j]);      \
        }                                                                                                              \
    }
#define PP_BARRIER(k_)                                                                                                 \
    {                                                                                                                  \
        if ((k_) + 1 < nk) PP_WAIT((k_) + 1)                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
    }
        const int last = nk - 1;
#ifdef KTF_TILE_PROBE
        long long pt[6] = {0, 0, 0, 0, 0, 0};
#define PP_T(i_) if (ks == 10) pt[i_] = clock64();
#else
#define PP_T(i_)
#endif
        if (grp == 0) {
            if (PP_DMA_ON && is_ks < nk) XS_STAGE()                        // own share of stage 2
            PP_READ(0)
            for (int ks = 0; ks < nk; ++ks) {
                PP_T(0)
                if (PP_MFMA_ON) PP_MFMA() else PP_SLEEP()
                PP_T(1)
                PP_BARRIER(ks)
                PP_T(2)
                if (PP_DMA_ON && is_ks < nk) XS_STAGE()                    // own share of stage ks + 3
                PP_T(3)
                PP_READ(ks < last ? ks + 1 : last)
                PP_T(4)
            }
        } else {
            for (int ks = 0; ks < nk; ++ks) {
                PP_T(0)
                if (PP_DMA_ON && is_ks < nk) XS_STAGE()                    // own share of stage ks + 2
                PP_T(1)
                PP_READ(ks)
                PP_T(2)
                PP_BARRIER(ks)
                PP_T(3)
                if (PP_MFMA_ON) PP_MFMA() else PP_SLEEP()
                PP_T(4)
            }
        }
#ifdef KTF_TILE_PROBE
        if (p.probe && (tid == 0 || tid == 256) && blockIdx.x < 4096) {
            long long* q = p.probe + (int64_t)(65536 + blockIdx.x * 2 + grp) * 8;       // behind the per-tile stamps
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = pt[i];
        }
#endif
#undef PP_T
#undef PP_READ
#undef PP_MFMA
#undef PP_BARRIER
#undef PP_ABL4
#undef PP_DMA_ON
#undef PP_MFMA_ON
#undef PP_SLEEP
#undef PP_VM
#undef PP_WAIT
    } else if constexpr (PIPE == 1) {
        if (KTF_X3_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);     // static priority for the later-dispatched half (A/B)
        const int doff = (wave >= 4) ? KTF_X3_DOFF : 0;        // wave-uniform
        for (int ks = 0; ks < nk; ++ks) {
            // stage ks landed: nothing else is in flight (two stages), or only the DMAs of stage ks+1 are (three stages)
            if (NST == 4 && ks + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // (one-pass form) stages ks + 1, ks + 2 in flight
            else if (NST >= 3 && ks + 1 < nk) {             // stage ks + 1 may stay in flight: four DMAs per thread, six with a residual plane
                if (TERMS == 1 || ks + 1 >= p.lo_steps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ks == 0) { XS_PROBE(2) }
            const unsigned char* sa = rsm + cur_slot * STG;
            const unsigned char* sw = sa + WOFF;
            cur_slot = (cur_slot + 1 == NST) ? 0 : cur_slot + 1;
            const bool refill = is_ks < nk;                     // next stage -> the buffer every wave finished reading
            unsigned char* st_ = rsm + fill_slot * STG + wave * 1024;
            // DMA n of the stage: 0,1 = A hi / lo rows 0-127; 2,3 = rows 128-255; 4,5 = W hi / lo rows 0-127; 6,7 = rows 128-255
#define XS_DMA(n)                                                                                                      \
    {                                                                                                                  \
        const char* src_ = ((n) < 4) ? ((((n) & 1) ? xl : xh) + va[(n) >> 1]) : ((((n) & 1) ? wl : wh) + vw[((n) - 4) >> 1]); \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)src_,                                                             \
            (lds_ptr_t*)(st_ + (((n) < 4) ? ((n) & 1) * R_TILE_BYTES : WOFF + ((n) & 1) * R_TILE_BYTES) + (((n) >> 1) & 1) * 8192), 16, 0, \
            ((n) < 4) ? KTF_X3_A_AUX : KTF_X3_W_AUX);                                                                  \
    }
            const bool two = TERMS != 2 || ks < p.lo_steps;              // this step has a weight residual (always, outside the 2-pass form)
            const bool two_next = TERMS != 2 || is_ks < p.lo_steps;     // ... and so has the stage being fetched
            bfrag8 bh[4], bl[4], af[2][4];                      // af[set][0,1] = hi fragments of the group's two rows, [2,3] = lo
            // fragment reads in the order the MFMAs consume them (LDS returns in order: the first MFMA waits for two reads, not twelve)
            af[0][0] = *reinterpret_cast<const bfrag8*>(sa + a_row_off);
            bh[0] = *reinterpret_cast<const bfrag8*>(sw + b_row_off);
            __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise moves the A read behind the eight B reads)
#pragma unroll
            for (int j = 1; j < 4; ++j) bh[j] = *reinterpret_cast<const bfrag8*>(sw + b_row_off + j * 16 * 64);
            __builtin_amdgcn_sched_barrier(0);
            if (TERMS == 3) af[0][2] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off);
            if (TERMS > 1 && two) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const bfrag8*>(sw + R_TILE_BYTES + b_row_off + j * 16 * 64);
            }
            af[0][1] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + 16 * 64);
            if (TERMS == 3) af[0][3] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off + 16 * 64);
            unsigned va[2], vw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int r_ = a_t[i] + is_off;
                r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);
                va[i] = (unsigned)r_ * x_rm + a_cb[i] + is_xb;
                vw[i] = w_ob[i] + (unsigned)is_ks * w_step;
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int PER_ROW = 4 * TERMS, PER_CHUNK = PER_ROW / 2;      // MFMAs per tile row / per chunk (4 chunks per 2-row group)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cur = g & 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {                  // chunk c = MFMAs 6c .. 6c+5 of the group's 24 (4c .. 4c+3 of 16)
                    // 2-pass form: the odd chunks are the residual passes of the group's two rows; a step behind the residual
                    // prefix skips them (wave-uniform)
                    if (TERMS != 2 || !(c & 1) || two)
#pragma unroll
                    for (int m = PER_CHUNK * c; m < PER_CHUNK * c + PER_CHUNK; ++m) {
                        const int r = m / PER_ROW, j = m & 3;                  // row, column block
                        const int t = (TERMS == 3) ? (m % PER_ROW) / 4 : 2 * ((m % PER_ROW) / 4);   // term: 0 hh, 1 lh, 2 hl
#if defined(KTF_X3_ABL) && (KTF_X3_ABL & 1)   // timing-only ablations (wrong results). 1: drop one of the three MFMA passes
                        if (t == 2) continue;
#endif
                        f32x4v& cc = acc[2 * g + r][j];
                        if constexpr (PK) cc = mfma16x16x32<F16>(t == 2 ? bl[j] : bh[j], af[cur][r], cc);
                        else cc = mfma16x16x32<F16>(t == 1 ? af[cur][2 + r] : af[cur][r], t == 2 ? bl[j] : bh[j], cc);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (c == 0 && g < 3) {
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            af[cur ^ 1][r] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + (2 * (g + 1) + r) * 16 * 64);
                            if (TERMS == 3) af[cur ^ 1][2 + r] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off + (2 * (g + 1) + r) * 16 * 64);
                        }
                    }
#if defined(KTF_X3_ABL) && (KTF_X3_ABL & 2)   // 2: no steady-state operand DMA (the MFMAs chew on stale stages)
                    if (false) {
#else
                    if (refill) {
#endif
                        const int n = 4 * g + c - doff;         // slot -> DMA index (wave-uniform)
                        if (TERMS == 3) {
                            if (n == 0) XS_DMA(0) else if (n == 1) XS_DMA(1) else if (n == 2) XS_DMA(2) else if (n == 3) XS_DMA(3)
                            else if (n == 4) XS_DMA(4) else if (n == 5) XS_DMA(5) else if (n == 6) XS_DMA(6) else if (n == 7) XS_DMA(7)
                        } else if (TERMS == 1) {                   // one pass: no residual plane at all, four DMAs
                            if (n == 0) XS_DMA(0) else if (n == 1) XS_DMA(2) else if (n == 2) XS_DMA(4) else if (n == 3) XS_DMA(6)
                        } else {                                   // no residual plane of the activations: six DMAs
#if defined(KTF_X3_ABL) && (KTF_X3_ABL & 64)    // 64: no steady-state DMA of the activations (what a shared A window would save, x 2/3)
                            if (n == 2) XS_DMA(4) else if (n == 3) XS_DMA(5)
#elif defined(KTF_X3_ABL) && (KTF_X3_ABL & 128)  // 128: no steady-state DMA of the weight residual plane
                            if (n == 0) XS_DMA(0) else if (n == 1) XS_DMA(2) else if (n == 2) XS_DMA(4)
                            else if (n == 4) XS_DMA(6)
#else
                            if (n == 0) XS_DMA(0) else if (n == 1) XS_DMA(2) else if (n == 2) XS_DMA(4) else if (n == 3) { if (two_next) XS_DMA(5) }
#endif
                            else if (n == 4) XS_DMA(6) else if (n == 5) { if (two_next) XS_DMA(7) }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef XS_DMA
            if (refill) {
                fill_slot = (fill_slot + 1 == NST) ? 0 : fill_slot + 1;
                ++is_ks;
                if (p.kinter) {                  // next context of the same 32 features; after the last one, the next features
                    if (++is_c == p.nctx) {
                        is_c = 0;
                        is_db += R_BK * 2;
                        is_xb += x_cs;
                    }
                    is_off = p.ctx[is_c];
                } else {
                    is_db += R_BK * 2;
                    is_xb += x_cs;
                    if (is_db == dpad_b) {
                        is_db = 0;
                        is_xb = 0;
                        ++is_c;
                        is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;
                    }
                }
            }
        }
    } else
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stage ks landed (nothing else is in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (is_ks < nk) XS_STAGE()                           // stage ks+1 -> the buffer every wave finished reading
        const unsigned char* sa = rsm + (ks & 1) * XS_STAGE_BYTES;
        const unsigned char* sw = sa + 2 * R_TILE_BYTES;
        bfrag8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = *reinterpret_cast<const bfrag8*>(sw + b_row_off + j * 16 * 64);
            bl[j] = *reinterpret_cast<const bfrag8*>(sw + R_TILE_BYTES + b_row_off + j * 16 * 64);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bfrag8 ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + (half * 4 + i) * 16 * 64);
                al[i] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off + (half * 4 + i) * 16 * 64);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4v(&c)[4] = acc[half * 4 + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = mfma16x16x32<false>(ah[i], bh[j], c[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = mfma16x16x32<false>(al[i], bh[j], c[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = mfma16x16x32<false>(ah[i], bl[j], c[j]);
            }
        }
    }
#undef XS_STAGE
    XS_PROBE(3)
    if (!STATS) __syncthreads();      // all fragment reads done before the LDS is reused as the store staging area
    if constexpr (PK) ring_epilogue16_pk<ACT, F16>(acc, p, rsm, b, t0, n0, out_len, wm, wn, wave, lane);
    else ring_epilogue16<ACT, STATS, F16>(acc, p, stats, rsm, b, t0, n0, out_len, wm, wn, wave, lane, eprm);
    XS_PROBE(4)
#undef XS_PROBE
}

// ------------------------------------------------------------------------------------ F32, few workgroups (latency)
// A single utterance (M = 998) gives the 128-tiles 32 workgroups and even 64-tiles only one workgroup per CU: nothing
// hides a global-load round trip, and a register-staged prefetch gets serialised by the compiler's vmcnt placement. These
// two kernels stage through an LDS ring filled by LDS-DMA with counted waits instead. Both accumulate in K order with
// fp32 FMAs -- the summation order of v_mfma_f32_32x32x2_f32 -- so they are bit-identical to the 128x128 tile kernel and a
// batch still equals its single-utterance calls.
//
// (1) 64 x BN tile, NB 16x16 blocks (v_mfma_f32_16x16x4_f32) per wave that share the A fragment: <BN, NB> = <32, 1> eight waves,
//     <64, 1> sixteen, <96, 3> eight (the 1500-unit layer of one utterance: 256 workgroups in ONE round instead of 384 in
//     two, 36 -> 21 us). K-step 64 when the per-context width
//     allows it, else 32; 4-stage LDS-DMA ring (up to 160 KiB), loads 4 steps ahead, one to five 16-byte DMAs per thread and
//     stage. Rows are BK*4 bytes; chunk c of row r sits at position c ^ (r & (CH-1)) (2-way on the scalar fragment reads).
//     Measured at K = 1536 on one utterance (998 x 512 outputs, 128 workgroups): 40 us; four waves of one 32x32x2 block
//     53 us (a dependent fp32 MFMA costs ~120 cycles against 64 of issue); four waves of 2x2 16x16x4 blocks 43 us; K-step
//     32 with this shape 44 us; 8 stages / 7 steps ahead the same. Timing-only ablations (K-step 32): without the refill
//     DMAs 41 us, without the MFMAs 24 us -- the step is the CU's fp32 MFMA time (64x64x32 = 1024 cycles) plus about as
//     much LDS fragment traffic (each operand block is read by four waves), which one workgroup per CU cannot overlap.
//     BN = 32 (64 x 32 tiles, eight waves) when 64 x 64 tiles would leave CUs idle: twice the workgroups, half the MFMA and
//     LDS time per CU for 1.5x the L2->LDS bytes (the same layer: 26 us).
#define FS_BM 64
#define FS_NSTAGE 4
#ifndef KTF_FS_NB32
#define KTF_FS_NB32 1  // 16-column blocks per wave of the 64 x 32 tile (2: four waves, one A fragment feeds two MFMAs)
#endif
#ifndef KTF_FS_ABL
#define KTF_FS_ABL 0   // timing-only ablations (tools/b1_tile_probe.py; results are garbage): 1 no refill DMAs, 2 no MFMAs,
#endif                 // 4 no fragment reads, 8 no barrier
template <int BK, int BN, int NB>
__global__ __launch_bounds__(64 * 4 * (BN / 16 / NB)) void tdnn_f32s_kernel(TdnnParams p) {
    static_assert(BN % (16 * NB) == 0, "a wave owns NB 16-column blocks");
    constexpr int WN = BN / 16 / NB;                         // waves across the tile's columns, NB blocks each (one A fragment
    constexpr int NT = 64 * 4 * WN;                          // feeds NB MFMAs); 1024 / 512 threads
    constexpr int CH = BK / 4;                               // 16-byte chunks per row
    constexpr int ROWB = BK * 4;                             // bytes per staged row
    constexpr int A_BYTES = FS_BM * ROWB, W_BYTES = BN * ROWB;
    constexpr int TILE_BYTES = A_BYTES;                      // offset of the W tile inside a stage
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr bool HALVES = (FS_BM * CH + BN * CH == NT);    // <32,64>: threads 0-511 stage A, 512-1023 stage W
    constexpr int NA = HALVES ? 1 : (FS_BM * CH) / NT;       // DMAs per thread and stage into the A tile
    constexpr int NW = HALVES ? 0 : (BN * CH) / NT;          // ... and into the W tile
    constexpr int NDMA = HALVES ? 1 : NA + NW;
    static_assert(HALVES || ((FS_BM * CH) % NT == 0 && (BN * CH) % NT == 0), "staging does not divide");
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
#ifdef KTF_TILE_PROBE
    long long* fprobe = p.probe ? p.probe + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#define FS_PROBE(k) if (fprobe && threadIdx.x == 0) { fprobe[k] = wall_clock64(); fprobe[4 + k] = clock64(); }
#else
#define FS_PROBE(k)
#endif
    FS_PROBE(0)
    const int b = blockIdx.z;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = blockIdx.y * FS_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // staging: chunk q of a tile -> row q / CH, LDS position q % CH holds global chunk (q % CH) ^ (row & (CH-1)); a DMA
    // instruction of the workgroup covers NT consecutive chunks
    const bool isw = HALVES && tid >= NT / 2;
    constexpr int NAq = NA > 0 ? NA : 1, NWq = NW > 0 ? NW : 1;
    int a_t[NAq];
    unsigned a_cb[NAq], w_ob[NWq];
#pragma unroll
    for (int i = 0; i < NAq; ++i) {
        const int q = HALVES ? (tid & (NT / 2 - 1)) : i * NT + tid;
        const int row = q / CH;
        a_cb[i] = (unsigned)(((q % CH) ^ (row & (CH - 1))) * 16);
        a_t[i] = start + (t0 + row) * p.sub;
    }
#pragma unroll
    for (int i = 0; i < NWq; ++i) {
        const int q = HALVES ? (tid & (NT / 2 - 1)) : i * NT + tid;
        const int row = q / CH;
        w_ob[i] = (unsigned)(n0 + row) * (unsigned)p.ktot * 4u + (unsigned)(((q % CH) ^ (row & (CH - 1))) * 16);
    }
    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * 4;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const unsigned ldxb = (unsigned)p.ldx * 4u;
    const int nk = p.ktot / BK;
    const int lenm1 = len - 1;
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 4;
    // A stage is fetched in three pieces so that its DMAs can be spread over a K-step: FS_SRC (source offsets of stage
    // is_ks into soff[], LDS destination st_d), FS_ADV (scalar cursor to the next stage; holds the only scalar load, of a
    // context offset), FS_DMA(i) (the i-th 16-byte-per-lane DMA of the stage).
    unsigned soff[NDMA];
    unsigned char* st_d;
#define FS_SRC()                                                                                                       \
    {                                                                                                                  \
        st_d = fsm + (is_ks & (FS_NSTAGE - 1)) * STAGE_BYTES + wave * 1024;                                            \
        if (HALVES) {                                        /* waves 8-15 land in the W tile: wave * 1024 >= A_BYTES */ \
            int r_ = a_t[0] + is_off;                                                                                  \
            r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                               \
            soff[0] = isw ? w_ob[0] + (unsigned)(is_ks * ROWB) : (unsigned)r_ * ldxb + a_cb[0] + (unsigned)is_db;      \
        } else {                                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
                int r_ = a_t[i] + is_off;                                                                              \
                r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                           \
                soff[i] = (unsigned)r_ * ldxb + a_cb[i] + (unsigned)is_db;                                             \
            }                                                                                                          \
            _Pragma("unroll") for (int i = 0; i < NW; ++i) soff[NA + i] = w_ob[i] + (unsigned)(is_ks * ROWB);          \
        }                                                                                                              \
    }
#define FS_DMA(i_)                                                                                                     \
    {                                                                                                                  \
        if (HALVES)                                                                                                    \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)((isw ? wb : xb) + soff[0]), (lds_ptr_t*)st_d, 16, 0, 0);     \
        else if ((i_) < NA)                                                                                            \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + soff[i_]), (lds_ptr_t*)(st_d + (i_) * (NT * 16)), 16, 0, 0); \
        else                                                                                                           \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + soff[i_]),                                              \
                                             (lds_ptr_t*)(st_d + TILE_BYTES + ((i_) - NA) * (NT * 16)), 16, 0, 0);     \
    }
#define FS_ADV()                                                                                                       \
    {                                                                                                                  \
        ++is_ks;                                                                                                       \
        is_db += ROWB;                                                                                                 \
        if (is_db == dpad_b) {                                                                                         \
            is_db = 0;                                                                                                 \
            ++is_c;                                                                                                    \
            is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                                \
        }                                                                                                              \
    }
#define FS_STAGE()                                                                                                     \
    {                                                                                                                  \
        FS_SRC()                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < NDMA; ++i) FS_DMA(i)                                                     \
        FS_ADV()                                                                                                       \
    }
    for (int s_ = 0; s_ < FS_NSTAGE && s_ < nk; ++s_) FS_STAGE()

    f32x4v acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.0f;
    const int r16 = lane & 15, kq = lane >> 4;
    // Fragments: lane (row r16, quarter kq) takes element 4c + kq of its row for position c -- one float of each 16-byte
    // chunk, a ds_read_b32 per operand and MFMA. (Tried: the lane reads chunk 4g + kq whole and the four lanes of a row
    // transpose their 4 x 4 floats with v_permlane32_swap / v_permlane16_swap -- a quarter of the LDS instructions, same
    // bits, 13 % slower: the swaps are slower than the reads they replace.)
    const int a_row_off = (wm * 16 + r16) * ROWB + kq * 4;
    const int b_row_off = TILE_BYTES + (wn * NB * 16 + r16) * ROWB + kq * 4;        // block j: + j * 16 rows
    const int sw = r16 & (CH - 1);
    // wait until at most `n_` (0..3) of this wave's stages are still in flight (NDMA DMAs each); s_waitcnt with vmcnt = v,
    // expcnt / lgkmcnt left at their maxima
#define FS_VM(v_) (((v_) & 15) | (((v_) >> 4) << 14) | 0x0f70)
#define FS_WAIT(n_)                                                                                                    \
    {                                                                                                                  \
        static_assert(3 * NDMA <= 63, "vmcnt range");                                                                  \
        const int n__ = (n_);                                                                                          \
        if (n__ >= 3) __builtin_amdgcn_s_waitcnt(FS_VM(3 * NDMA));                                                     \
        else if (n__ == 2) __builtin_amdgcn_s_waitcnt(FS_VM(2 * NDMA));                                                \
        else if (n__ == 1) __builtin_amdgcn_s_waitcnt(FS_VM(NDMA));                                                    \
        else __builtin_amdgcn_s_waitcnt(FS_VM(0));                                                                     \
    }
    // The fragments of step ks + 1 are read under the MFMAs of step ks (one workgroup per CU, both waves of a SIMD in the
    // same phase: read latency in front of the MFMAs was 40 % of the step). A stage is refilled four steps ahead, into
    // the slot whose fragments every wave took during the previous step.
    float av[CH], bv[NB][CH];
    {
        const int issued = nk < FS_NSTAGE ? nk : FS_NSTAGE;
        FS_WAIT(issued - 1)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        FS_PROBE(1)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            av[c] = *reinterpret_cast<const float*>(fsm + a_row_off + ((c ^ sw) << 4));
#pragma unroll
            for (int j = 0; j < NB; ++j)
                bv[j][c] = *reinterpret_cast<const float*>(fsm + b_row_off + j * 16 * ROWB + ((c ^ sw) << 4));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as an instruction the compiler's counter model sees
    }                                                        // (an inline-asm wait is not: fragments "pending" at the loop head
                                                             // put a wait for the reads just issued in front of every MFMA)
    for (int ks = 0; ks + 1 < nk; ++ks) {
        const int beyond = nk - 2 - ks;                      // stages issued beyond ks + 1: min(beyond, 2)
        FS_WAIT(beyond < 2 ? beyond : 2)
        if (!(KTF_FS_ABL & 8)) __builtin_amdgcn_s_barrier(); // every wave has taken stage ks (its reads were waited for at the
        asm volatile("" ::: "memory");                       // end of the previous step): the slot can be refilled
        const bool refill = !(KTF_FS_ABL & 1) && is_ks < nk;
        if (refill) {
            FS_SRC()
            FS_ADV()
        }
        // One position of the K-step at a time: its MFMA(s), then the fragment reads of the same position of the next stage
        // into the registers those MFMAs just consumed, and every CH / NDMA positions one DMA of the refill. Bursts keep all
        // waves in LDS issue (at most 15 LDS operations of a wave are in flight) or in the texture addresser's queue while
        // the matrix pipes idle. In-kernel stamps (K = 1536, 64 x 32 tiles, tools/b1_tile_probe.py with -DKTF_FS_ABL): K-loop
        // 22.7 us; MFMAs + barrier alone 13.8, fragment reads + barrier alone 13.9 (256 ds_read_b32 per step and workgroup
        // at ~4.8 cycles each), DMA stream alone 9.4: the LDS instruction rate and the MFMAs are both near their limits.
        const unsigned char* nst = fsm + ((ks + 1) & (FS_NSTAGE - 1)) * STAGE_BYTES;
        constexpr int DSTEP = CH / NDMA > 0 ? CH / NDMA : 1;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (KTF_FS_ABL & 2) acc[j][c & 3] += av[c] * bv[j][c];
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[j][c], acc[j], 0, 0, 0);
            }
            if (c % DSTEP == 0 && c / DSTEP < NDMA) {
                if (refill) FS_DMA(c / DSTEP)
            }
            if (KTF_FS_ABL & 4) {
                av[c] += 1.0f;
            } else {
                av[c] = *reinterpret_cast<const float*>(nst + a_row_off + ((c ^ sw) << 4));
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    bv[j][c] = *reinterpret_cast<const float*>(nst + b_row_off + j * 16 * ROWB + ((c ^ sw) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[j][c], acc[j], 0, 0, 0);
#undef FS_WAIT
#undef FS_VM
    FS_PROBE(2)
#undef FS_STAGE
#undef FS_SRC
#undef FS_DMA
#undef FS_ADV
    // 16x16 accumulator layout: acc[r] = out[row 4*(lane>>4) + r][col lane&15]
    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + (wn * NB + j) * 16 + r16;
        if (n >= p.units) continue;
        const float bias = p.bias ? p.bias[n] : 0.0f;
        const float sc = p.scale ? p.scale[n] : 1.0f;
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = wm * 16 + kq * 4 + r;
            if (m < rows_valid) {
                float v = apply_act(acc[j][r] + bias, p.act);
                if (p.scale) v = v * sc + sh;
                const int64_t off = (out_row0 + m) * p.ldy + n;
                if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off] = v;
                else reinterpret_cast<unsigned short*>(p.y)[off] = f2bf(v);
            }
        }
    }
    FS_PROBE(3)
#undef FS_PROBE
}

// (3) throughput form: 128x128 tile, EIGHT waves of 2x4 blocks of v_mfma_f32_16x16x4_f32 (six scalar LDS reads feed eight
//     MFMAs), K-step 32, double-buffered LDS-DMA stages (64 KiB) and <= 128 VGPRs, so TWO workgroups share a CU and one's
//     prologue / epilogue / stage wait overlaps the other's MFMAs (in-kernel stamps on the register-staged 32x32x2 kernel:
//     K-loop 200-250 us with three workgroups per CU taking turns, then 80-130 us of epilogue per tile). Operands swapped
//     (W block as A): a lane owns four consecutive output columns of one row and stores 16 bytes. Same K order, same bits.
//     122 TFLOP/s at B = 1024 against 111 for the register-staged kernel (K-step 16 with four workgroups per CU: the same).
#define FT_BM 128
#define FT_BK 32
#define FT_TILE_BYTES (FT_BM * FT_BK * 4)            // 16 KiB per operand
#define FT_STAGE_BYTES (2 * FT_TILE_BYTES)
#define FT_LDS_BYTES (2 * FT_STAGE_BYTES)            // 64 KiB
template <int ACT>
__device__ __forceinline__ void f32t_epilogue(f32x4v (&acc)[2][4], const TdnnParams& p, int b, int t0, int n0, int out_len,
                                              int wm, int wn, int lane) {
    const int r16 = lane & 15, kq = lane >> 4;
    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    const bool vec_ok = (p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + kq * 4;
        float bias[4], sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool nv = n + e < p.units;
            bias[e] = (nv && p.bias) ? p.bias[n + e] : 0.0f;
            sc[e] = (nv && p.scale) ? p.scale[n + e] : 1.0f;
            sh[e] = (nv && p.shift) ? p.shift[n + e] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = wm * 32 + i * 16 + r16;
            if (m >= rows_valid) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = apply_act(acc[i][j][e] + bias[e], ACT);
                if (p.scale) v[e] = v[e] * sc[e] + sh[e];
            }
            const int64_t off = (out_row0 + m) * p.ldy + n;
            if (p.y_dtype == KTF_F32) {
                float* yp = reinterpret_cast<float*>(p.y) + off;
                if (vec_ok && n + 4 <= p.units) {
                    *reinterpret_cast<fv4*>(yp) = fv4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.units) yp[e] = v[e];
                }
            } else {
                unsigned short* yp = reinterpret_cast<unsigned short*>(p.y) + off;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.units) yp[e] = f2bf(v[e]);
            }
        }
    }
}

template <int BK>
__global__ __launch_bounds__(512, 2) void tdnn_f32t_kernel(TdnnParams p) {
    constexpr int CH = BK / 4;                               // 16-byte chunks per row
    constexpr int ROWB = BK * 4;
    constexpr int TILE_BYTES = FT_BM * ROWB;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int NDMA = (FT_BM * CH) / 512;                 // DMAs per thread, stage and operand: 2 (BK 32) / 1 (BK 16)
    extern __shared__ __attribute__((aligned(16))) unsigned char ftm[];
    const int b = blockIdx.z;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    if (p.out_lens && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
    const int t0 = blockIdx.y * FT_BM;
    if (t0 >= out_len || len <= 0) return;
    const int n0 = blockIdx.x * FT_BM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // staging: chunk q = i*512 + tid of a tile -> row q/CH, LDS position q%CH holds global chunk (q%CH) ^ (row&(CH-1))
    int a_t[NDMA];
    unsigned a_cb[NDMA], w_ob[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int q = i * 512 + tid;
        const int row = q / CH;
        const unsigned chunk = (unsigned)(((q % CH) ^ (row & (CH - 1))) * 16);
        a_cb[i] = chunk;
        a_t[i] = start + (t0 + row) * p.sub;
        w_ob[i] = (unsigned)(n0 + row) * (unsigned)p.ktot * 4u + chunk;
    }
    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * 4;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const unsigned ldxb = (unsigned)p.ldx * 4u;
    const int nk = p.ktot / BK;
    const int lenm1 = len - 1;
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 4;
#define FT_STAGE()                                                                                                     \
    {                                                                                                                  \
        unsigned char* st_ = ftm + (is_ks & 1) * STAGE_BYTES + wave * 1024;                                            \
        _Pragma("unroll") for (int i = 0; i < NDMA; ++i) {                                                             \
            int r_ = a_t[i] + is_off;                                                                                  \
            r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                               \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + ((unsigned)r_ * ldxb + a_cb[i] + (unsigned)is_db)),    \
                                             (lds_ptr_t*)(st_ + i * 8192), 16, 0, 0);                                  \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + (w_ob[i] + (unsigned)(is_ks * ROWB))),                  \
                                             (lds_ptr_t*)(st_ + TILE_BYTES + i * 8192), 16, 0, 0);                     \
        }                                                                                                              \
        ++is_ks;                                                                                                       \
        is_db += ROWB;                                                                                                 \
        if (is_db == dpad_b) {                                                                                         \
            is_db = 0;                                                                                                 \
            ++is_c;                                                                                                    \
            is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                                \
        }                                                                                                              \
    }
    FT_STAGE()

    f32x4v acc[2][4];                                        // [row block i][column block j] of the wave's 32 x 64 outputs
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
    const int r16 = lane & 15, kq = lane >> 4;
    const int sw = r16 & (CH - 1);
    const int a_row_off = (wm * 32 + r16) * ROWB + kq * 4;
    const int b_row_off = TILE_BYTES + (wn * 64 + r16) * ROWB + kq * 4;
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stage ks landed (nothing else is in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (is_ks < nk) FT_STAGE()                           // stage ks+1 -> the buffer every wave finished reading
        const unsigned char* st = ftm + (ks & 1) * STAGE_BYTES;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int co = (c ^ sw) << 4;                    // rows r, r+16, r+32, r+48 share r & (CH-1): same position
            float av[2], bv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const float*>(st + a_row_off + i * 16 * ROWB + co);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float*>(st + b_row_off + j * 16 * ROWB + co);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);
        }
    }
#undef FT_STAGE
    if (p.act == KTF_ACT_NONE) f32t_epilogue<KTF_ACT_NONE>(acc, p, b, t0, n0, out_len, wm, wn, lane);
    else if (p.act == KTF_ACT_RELU) f32t_epilogue<KTF_ACT_RELU>(acc, p, b, t0, n0, out_len, wm, wn, lane);
    else if (p.act == KTF_ACT_SIGMOID) f32t_epilogue<KTF_ACT_SIGMOID>(acc, p, b, t0, n0, out_len, wm, wn, lane);
    else f32t_epilogue<KTF_ACT_TANH>(acc, p, b, t0, n0, out_len, wm, wn, lane);
}

// (2) <= 8 output rows in all (tdnn6 of a single utterance: one 3000-long row against 512 units; the 64-tiles would run 8
//     workgroups through a 94-step serial loop). One single-wave workgroup owns 16 units of ONE output row: all lanes
//     issue the DMAs of a 16 x 32 weight slice and the row's 32 inputs into a 16-deep ring (loads 14 steps ahead, no
//     barrier: one wave), lanes 0-15 run the fmaf chain.
#define RV_UNITS 16
#define RV_BK 32
#define RV_NSTAGE 16
#define RV_STAGE_BYTES (RV_UNITS * RV_BK * 4 + 256)            // 2 KiB of W + the row's 32 inputs (a 4-byte DMA writes 64 lanes x 4 B: stored twice)
__global__ __launch_bounds__(64) void tdnn_f32_rowvec_kernel(TdnnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char rvm[RV_NSTAGE * RV_STAGE_BYTES];
    const int b = blockIdx.z, t = blockIdx.y;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    int start;
    const int out_len = tdnn_out_len(len, p, start);
    const int tid = threadIdx.x;
    if (p.out_lens && blockIdx.x == 0 && t == 0 && tid == 0) p.out_lens[b] = out_len;
    if (t >= out_len || len <= 0) return;
    const int n0 = blockIdx.x * RV_UNITS;
    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * 4;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const unsigned ldxb = (unsigned)p.ldx * 4u;
    const int at = start + t * p.sub;
    unsigned w_ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 64 + tid;
        const int row = q >> 3;
        w_ob[i] = (unsigned)(n0 + row) * (unsigned)p.ktot * 4u + (unsigned)(((q & 7) ^ ((row >> 1) & 7)) * 16);
    }
    const int nk = p.ktot / RV_BK;
    const int lenm1 = len - 1;
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 4;
#define RV_STAGE()                                                                                                     \
    {                                                                                                                  \
        unsigned char* st_ = rvm + (is_ks & (RV_NSTAGE - 1)) * RV_STAGE_BYTES;                                         \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + (w_ob[0] + (unsigned)(is_ks * (RV_BK * 4)))), (lds_ptr_t*)(st_), 16, 0, 0);        \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb + (w_ob[1] + (unsigned)(is_ks * (RV_BK * 4)))), (lds_ptr_t*)(st_ + 1024), 16, 0, 0); \
        int r_ = at + is_off;                                                                                          \
        r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                                   \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + ((unsigned)r_ * ldxb + (unsigned)is_db + (unsigned)((tid & 31) * 4))), \
                                         (lds_ptr_t*)(st_ + 2048), 4, 0, 0);                                           \
        ++is_ks;                                                                                                       \
        is_db += RV_BK * 4;                                                                                            \
        if (is_db == dpad_b) {                                                                                         \
            is_db = 0;                                                                                                 \
            ++is_c;                                                                                                    \
            is_off = (is_c < p.nctx) ? p.ctx[is_c] : 0;                                                                \
        }                                                                                                              \
    }
    for (int s_ = 0; s_ < RV_NSTAGE && s_ < nk; ++s_) RV_STAGE()       // all sixteen slots
    float acc = 0.0f;
    const int u = tid & 15;
    const int sw = (u >> 1) & 7;
    // The chain of K dependent FMAs is the floor (its order is the batch kernels' order). The 16 fragment reads of step
    // ks + 1 are issued BEFORE the 32 FMAs of step ks (two register sets, loop unrolled by two so that no set is copied):
    // with read -> wait -> FMA per step a third of the step was exposed LDS latency. Waits are s_waitcnt instructions the
    // compiler's counter model sees (behind an inline-asm wait it re-waits for the reads just issued in front of the FMAs).
#define RV_VM(v_) (((v_) & 15) | (((v_) >> 4) << 14) | 0x0f70)
    // stage j_ has landed: stages up to min(j_ + 14, nk - 1) have been issued, 3 DMAs each, completing in order
#define RV_LANDED(j_)                                                                                                  \
    {                                                                                                                  \
        if ((j_) + RV_NSTAGE - 2 <= nk - 1) __builtin_amdgcn_s_waitcnt(RV_VM(3 * (RV_NSTAGE - 2)));                    \
        else __builtin_amdgcn_s_waitcnt(RV_VM(0));                                                                     \
    }
#define RV_READ(wv_, xv_, j_)                                                                                          \
    {                                                                                                                  \
        const unsigned char* st_ = rvm + ((j_) & (RV_NSTAGE - 1)) * RV_STAGE_BYTES;                                    \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                \
            wv_[c] = *reinterpret_cast<const fv4*>(st_ + u * 128 + ((c ^ sw) << 4));                                   \
            xv_[c] = *reinterpret_cast<const fv4*>(st_ + 2048 + c * 16);                                               \
        }                                                                                                              \
    }
#define RV_FMA(wv_, xv_)                                                                                               \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                    \
        acc = fmaf(xv_[c].x, wv_[c].x, acc);                                                                           \
        acc = fmaf(xv_[c].y, wv_[c].y, acc);                                                                           \
        acc = fmaf(xv_[c].z, wv_[c].z, acc);                                                                           \
        acc = fmaf(xv_[c].w, wv_[c].w, acc);                                                                           \
    }
    // one step: reads of stage j_ + 1 into the OTHER set, FMAs of stage j_ from THIS set, then slot j_ (read one step ago) is refilled
#define RV_STEP(w_, x_, wn_, xn_, j_)                                                                                  \
    {                                                                                                                  \
        /* unconditional (the last step re-reads its own stage): a register set defined on one side of a branch only   \
           is a phi, and the compiler then parks a wait for the reads in front of the FMAs */                           \
        const int jn_ = (j_) + 1 < nk ? (j_) + 1 : nk - 1;                                                             \
        RV_LANDED(jn_)                                                                                                 \
        RV_READ(wn_, xn_, jn_)                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        RV_FMA(w_, x_)                                                                                                 \
        asm volatile("" : "+v"(acc));           /* the chain is complete HERE: without this the compiler sinks it below   \
                                                   the refill block, i.e. behind a wait for the reads just issued */     \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_waitcnt(0xc07f);                                                                            \
        if (is_ks < nk) RV_STAGE()                                                                                     \
    }
    fv4 w0[8], x0[8], w1[8], x1[8];
    RV_LANDED(0)
    RV_READ(w0, x0, 0)
    __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
    int ks = 0;
    for (; ks + 1 < nk; ks += 2) {
        RV_STEP(w0, x0, w1, x1, ks)
        RV_STEP(w1, x1, w0, x0, ks + 1)
    }
    if (ks < nk) RV_FMA(w0, x0)                              // odd step count: the last stage sits in set 0
#undef RV_VM
#undef RV_LANDED
#undef RV_READ
#undef RV_FMA
#undef RV_STEP
#undef RV_STAGE
    const int n = n0 + tid;
    if (tid < RV_UNITS && n < p.units) {
        const float bias = p.bias ? p.bias[n] : 0.0f;
        const float sc = p.scale ? p.scale[n] : 1.0f;
        const float sh = p.shift ? p.shift[n] : 0.0f;
        float v = apply_act(acc + bias, p.act);
        if (p.scale) v = v * sc + sh;
        const int64_t off = ((int64_t)b * p.Tout + t) * p.ldy + n;
        if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off] = v;
        else reinterpret_cast<unsigned short*>(p.y)[off] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------ elementwise helpers
__global__ void affine_act_kernel(const float* __restrict__ x, int64_t total, int D, int act,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  float* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        float v = apply_act(x[e], act);
        if (scale) v *= scale[d];
        if (shift) v += shift[d];
        y[e] = v;
    }
}

template <typename S>
__device__ __forceinline__ float cp_load(const S* p);
template <> __device__ __forceinline__ float cp_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float cp_load<unsigned short>(const unsigned short* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float cp_load<_Float16>(const _Float16* p) { return (float)*p; }
template <typename Dd>
__device__ __forceinline__ void cp_store(Dd* p, float v);
template <> __device__ __forceinline__ void cp_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void cp_store<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void cp_store<_Float16>(_Float16* p, float v) { *p = (_Float16)v; }

template <typename S, typename Dd>
__global__ void convert_pad_kernel(const S* __restrict__ src, int64_t rows, int D, int64_t lds_, Dd* __restrict__ dst,
                                   int64_t ldd) {
    const int64_t total = rows * ldd;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / ldd;
        const int d = (int)(e - r * ldd);
        cp_store<Dd>(dst + e, d < D ? cp_load<S>(src + r * lds_ + d) : 0.0f);
    }
}

extern "C" int64_t ktf_tdnn_out_len(int64_t len, const KtfTdnnDesc* d) {
    if (!d || d->nctx <= 0 || d->subsampling <= 0) return -1;
    int64_t start = 0, end = len;
    if (d->valid) {
        if (d->ctx[0] < 0) start = -d->ctx[0];
        if (d->ctx[d->nctx - 1] > 0) end = len - d->ctx[d->nctx - 1];
    }
    const int64_t n = end - start;
    return n <= 0 ? 0 : (n + d->subsampling - 1) / d->subsampling;
}

static int tdnn_launch(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
                       const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift,
                       void* y, int64_t ldy, int32_t* out_lens, double* stats_sums, void* stream,
                       const void* x_lo = nullptr, void* y_lo = nullptr) {
    KTF_REQUIRE(x && d && w && (y || stats_sums), "ktf_tdnn: null argument");
    const bool split_in = d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16;     // activations as hi/lo bf16 planes
    const bool half2 = d->gemm == KTF_GEMM_F16X2;                                    // one half plane in, hi + lo half weights
    if (split_in) KTF_REQUIRE(x_lo, "ktf_tdnn_split: null lo plane");
    if (half2) {
        KTF_REQUIRE(d->x_dtype == KTF_F16 && d->w_dtype == KTF_F16 && !x_lo && !y_lo,
                    "ktf_tdnn: F16X2 takes ONE half activation plane (x_lo, y_lo NULL) and half weights as w (hi) + w_lo (w_lo NULL: one pass)");
        KTF_REQUIRE(d->units > 128 && (stats_sums || (ldy % 8 == 0 && (d->y_dtype == KTF_F16 || d->y_dtype == KTF_F32))),
                    "ktf_tdnn: F16X2 runs on the 256x256 kernel only (units > 128, ldy %% 8 == 0, half or fp32 output)");
    }
    if (y_lo) KTF_REQUIRE(split_in && d->y_dtype == KTF_BF16, "ktf_tdnn_split: a split output needs split input and y_dtype bf16");
    if (stats_sums) {
        KTF_REQUIRE(((d->gemm == KTF_GEMM_BF16 && d->x_dtype == KTF_BF16) || (d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_F32) || split_in || half2 ||
                     (d->gemm == KTF_GEMM_F16 && d->x_dtype == KTF_F16)) &&
                        d->units > 128 && !d->valid && d->subsampling == 1,
                    "ktf_tdnn_stats: needs a ring kernel (bf16, f16 or bf16x3 gemm, units > 128, SAME padding, no subsampling)");
        ldy = (d->units + 3) / 4 * 4;
    }
    KTF_REQUIRE(B >= 0 && T >= 0, "ktf_tdnn: negative size");
    KTF_REQUIRE(d->units > 0 && d->din > 0, "ktf_tdnn: units/din must be > 0");
    KTF_REQUIRE(d->nctx >= 1 && d->nctx <= 16, "ktf_tdnn: nctx %d outside [1,16]", d->nctx);
    for (int i = 1; i < d->nctx; ++i) KTF_REQUIRE(d->ctx[i] > d->ctx[i - 1], "ktf_tdnn: context must be strictly ascending");
    KTF_REQUIRE(d->subsampling > 0, "ktf_tdnn: subsampling_factor should be > 0");
    KTF_REQUIRE(d->din_pad >= d->din && d->din_pad % 32 == 0 && d->din_pad <= ldx, "ktf_tdnn: din_pad %d must be a multiple of 32 with din <= din_pad <= ldx", d->din_pad);
    KTF_REQUIRE(ldx % 8 == 0, "ktf_tdnn: ldx must be a multiple of 8");
    KTF_REQUIRE(ldy >= d->units, "ktf_tdnn: ldy < units");
    KTF_REQUIRE(d->act >= KTF_ACT_NONE && d->act <= KTF_ACT_TANH, "ktf_tdnn: bad activation %d", d->act);
    KTF_REQUIRE(d->y_dtype == KTF_F32 || d->y_dtype == KTF_BF16 || d->y_dtype == KTF_F16, "ktf_tdnn: bad y_dtype");
    KTF_REQUIRE(d->y_dtype != KTF_F16 || d->gemm == KTF_GEMM_F16 || half2, "ktf_tdnn: half output needs KTF_GEMM_F16 or KTF_GEMM_F16X2");
    KTF_REQUIRE((scale == nullptr) == (shift == nullptr), "ktf_tdnn: scale and shift go together");
    KTF_REQUIRE(T < (1ll << 30) && B < 65536, "ktf_tdnn: T or B too large");
    const int64_t Tout = ktf_tdnn_out_len(T, d);
    if (B == 0 || T == 0) return KTF_OK;
    if (Tout == 0) {
        if (out_lens) (void)hipMemsetAsync(out_lens, 0, sizeof(int32_t) * B, (hipStream_t)stream);
        return KTF_OK;
    }
    TdnnParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.lens = lens; p.w = w; p.w_lo = w_lo; p.bias = bias; p.scale = scale; p.shift = shift; p.y = y;
    p.x_lo = x_lo; p.y_lo = y_lo;
    p.out_lens = out_lens; p.T = T; p.ldx = ldx; p.ldy = ldy; p.Tout = Tout;
    p.units = d->units; p.din_pad = d->din_pad; p.nctx = d->nctx; p.sub = d->subsampling; p.valid = d->valid;
    p.act = d->act; p.y_dtype = d->y_dtype; p.ktot = d->nctx * d->din_pad;
    p.stat_slots = (stats_sums && (d->flags & KTF_TDNN_DET_STATS)) ? (int32_t)ktf_stats_slots(Tout) : 0;
    {
        const int pre = (d->flags >> 8) & 0xffff;             // KTF_TDNN_LO_PREFIX(chunks) = (chunks + 1) << 8
        p.lo_steps = pre ? (pre - 1) * d->nctx : INT32_MAX;
        if (pre) KTF_REQUIRE(d->gemm == KTF_GEMM_F16X2 && (d->flags & KTF_TDNN_K_INTERLEAVED) && w_lo,
                             "ktf_tdnn: KTF_TDNN_LO_PREFIX needs KTF_GEMM_F16X2 with K-interleaved weights and a residual plane");
    }
    p.kinter = (d->flags & KTF_TDNN_K_INTERLEAVED) ? 1 : 0;
    p.wtiled = (d->flags & KTF_TDNN_W_TILED) ? 1 : 0;
    p.xchunk = (d->flags & KTF_TDNN_X_CHUNKED) ? 1 : 0;
    p.ychunk = (d->flags & KTF_TDNN_Y_CHUNKED) ? 1 : 0;
    if (p.xchunk || p.ychunk) {
        KTF_REQUIRE(half2 || (split_in && d->units > 128), "ktf_tdnn: chunk-major activations are implemented by the split-plane kernel only");
        KTF_REQUIRE(!p.xchunk || ldx == d->din_pad, "ktf_tdnn: KTF_TDNN_X_CHUNKED needs ldx == din_pad (whole 32-feature chunks)");
        KTF_REQUIRE(!p.ychunk || (!stats_sums && ldy % 32 == 0 && d->y_dtype != KTF_F32), "ktf_tdnn: KTF_TDNN_Y_CHUNKED needs a 16-bit output with ldy %% 32 == 0");
    }
    if (p.wtiled) KTF_REQUIRE(half2 || (split_in && d->units > 128 && ldy % 4 == 0), "ktf_tdnn: KTF_TDNN_W_TILED is implemented by the split-plane kernel only");
#ifdef KTF_TILE_PROBE
    p.probe = KTF_PROBE_BUF;
#endif
    if (p.kinter) KTF_REQUIRE(half2 || (split_in && d->units > 128 && ldy % 4 == 0), "ktf_tdnn: KTF_TDNN_K_INTERLEAVED is implemented by the split-plane kernel only (ktf_tdnn_split*, units > 128)");
    for (int i = 0; i < d->nctx; ++i) p.ctx[i] = d->ctx[i];
    hipStream_t st = (hipStream_t)stream;
    const unsigned ntiles = (unsigned)ktf_cdiv(d->units, 128);
    if (d->gemm == KTF_GEMM_F32) {
        KTF_REQUIRE(d->x_dtype == KTF_F32 && d->w_dtype == KTF_F32, "ktf_tdnn: F32 gemm needs fp32 x and w");
        // W must cover round_up(units, 128) rows (the host pads to 256)
        const int64_t wg128 = (int64_t)ktf_cdiv(d->units, 128) * ktf_cdiv(Tout, 128) * B;
        const bool lat = !(d->flags & KTF_TDNN_REF_TILES);   // flag: the register-staged 32x32x2 tile kernels (bitwise reference of the DMA-staged ones)
        if (lat && B * Tout <= 8) {
            dim3 grid((unsigned)ktf_cdiv(d->units, RV_UNITS), (unsigned)Tout, (unsigned)B);
            hipLaunchKernelGGL(tdnn_f32_rowvec_kernel, grid, dim3(64), 0, st, p);
        } else if (lat && wg128 < 256) {
            const int64_t wg64 = (int64_t)ktf_cdiv(d->units, FS_BM) * ktf_cdiv(Tout, FS_BM) * B;
#define FS_LAUNCH(BK_, BN_, NB_)                                                                                       \
    do {                                                                                                               \
        const int lds = FS_NSTAGE * (FS_BM + BN_) * BK_ * 4;                                                           \
        dim3 grid_((unsigned)ktf_cdiv(d->units, BN_), (unsigned)ktf_cdiv(Tout, FS_BM), (unsigned)B);                   \
        KTF_LDS_ONCE(lds, tdnn_f32s_kernel<BK_, BN_, NB_>);                                                            \
        hipLaunchKernelGGL((tdnn_f32s_kernel<BK_, BN_, NB_>), grid_, dim3(64 * 4 * (BN_ / 16 / NB_)), lds, st, p);      \
    } while (0)
            if (d->din_pad % 64 == 0) {
                // tile width: one workgroup per CU (the ring takes most of the LDS), so the cost is (rounds of 256 workgroups) x
                // (time of one, ~ width + fixed part); 96 columns only where the padded W rows cover the last tile
                const int64_t mt = (int64_t)ktf_cdiv(Tout, FS_BM) * B;
                int best = 32;
                int64_t best_cost = INT64_MAX;
                for (int bn = 32; bn <= 96; bn += 32) {
                    if (bn == 96 && (int64_t)ktf_cdiv(d->units, 96) * 96 > (int64_t)ktf_cdiv(d->units, 128) * 128) continue;
                    const int64_t cost = ktf_cdiv(ktf_cdiv(d->units, bn) * mt, 256) * (bn + 16);
                    if (cost < best_cost) best_cost = cost, best = bn;
                }
                (void)wg64;
                if (best == 32) FS_LAUNCH(64, 32, KTF_FS_NB32);
                else if (best == 64) FS_LAUNCH(64, 64, 1);
                else FS_LAUNCH(64, 96, 3);
            } else {
                FS_LAUNCH(32, 64, 1);
            }
#undef FS_LAUNCH
        } else if (lat) {
            dim3 grid((unsigned)ktf_cdiv(d->units, FT_BM), (unsigned)ktf_cdiv(Tout, FT_BM), (unsigned)B);
            KTF_LDS_ONCE(FT_LDS_BYTES, tdnn_f32t_kernel<32>);
            hipLaunchKernelGGL(tdnn_f32t_kernel<32>, grid, dim3(512), FT_LDS_BYTES, st, p);
        } else if (wg128 >= 256) {
            dim3 grid((unsigned)ktf_cdiv(d->units, 128), (unsigned)ktf_cdiv(Tout, 128), (unsigned)B);
            hipLaunchKernelGGL((tdnn_f32_kernel<2, 16>), grid, dim3(256), 0, st, p);
        } else {
            dim3 grid((unsigned)ktf_cdiv(d->units, 64), (unsigned)ktf_cdiv(Tout, 64), (unsigned)B);
            hipLaunchKernelGGL((tdnn_f32_kernel<1, 32>), grid, dim3(256), 0, st, p);
        }
    } else if (half2) {
        const int mtiles = ktf_cdiv(Tout, R_BM), ntiles_r = ktf_cdiv(d->units, R_BN);
        const int64_t gtiles = B * (int64_t)mtiles;
        const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles_r;
        KTF_REQUIRE(nblocks < (1ll << 31), "ktf_tdnn: grid too large");
        KTF_REQUIRE(d->act == KTF_ACT_NONE || d->act == KTF_ACT_RELU, "ktf_tdnn: F16X2 fuses ReLU or no activation");
#define H2_LAUNCH(A, ST, PKD)                                                                                          \
    do {                                                                                                               \
        constexpr int pipe_ = (PKD) ? 1 : KTF_X2_PIPE;                                                                 \
        constexpr int lds_ = (PKD) ? (R16_LDS_BYTES > XS_LDS_BYTES ? R16_LDS_BYTES : XS_LDS_BYTES)                     \
                                   : (pipe_ == 2 ? 9 * R_TILE_BYTES : X2_LDS_BYTES);                                   \
        if (!w_lo && !(PKD)) {                               /* no residual plane: ONE pass */                          \
            constexpr int lds1_ = KTF_X1_STAGES * 2 * R_TILE_BYTES > 5 * R_TILE_BYTES ? KTF_X1_STAGES * 2 * R_TILE_BYTES : 5 * R_TILE_BYTES;   /* >= the epilogue's staging image */ \
            KTF_LDS_ONCE(lds1_, tdnn_x3s_kernel<A, ST, 1, true, 1, false>);                                            \
            hipLaunchKernelGGL((tdnn_x3s_kernel<A, ST, 1, true, 1, false>), dim3((unsigned)nblocks), dim3(512), lds1_, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(lds_, tdnn_x3s_kernel<A, ST, pipe_, true, 2, PKD>);                                           \
            hipLaunchKernelGGL((tdnn_x3s_kernel<A, ST, pipe_, true, 2, PKD>), dim3((unsigned)nblocks), dim3(512), lds_, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        }                                                                                                              \
    } while (0)
        const bool pk = !stats_sums && d->y_dtype == KTF_F16 && KTF_X2_PK;      // one half plane out: packed single-barrier epilogue
        if (d->act == KTF_ACT_RELU) { if (stats_sums) H2_LAUNCH(KTF_ACT_RELU, true, false); else if (pk) H2_LAUNCH(KTF_ACT_RELU, false, true); else H2_LAUNCH(KTF_ACT_RELU, false, false); }
        else { if (stats_sums) H2_LAUNCH(KTF_ACT_NONE, true, false); else if (pk) H2_LAUNCH(KTF_ACT_NONE, false, true); else H2_LAUNCH(KTF_ACT_NONE, false, false); }
#undef H2_LAUNCH
    } else if (d->gemm == KTF_GEMM_BF16 || d->gemm == KTF_GEMM_BF16X3 || d->gemm == KTF_GEMM_F16) {
        const bool f16 = d->gemm == KTF_GEMM_F16;
        if (f16) {
            KTF_REQUIRE(d->w_dtype == KTF_F16 && d->x_dtype == KTF_F16, "ktf_tdnn: F16 gemm needs half x and w");
            KTF_REQUIRE(d->y_dtype == KTF_F16 || d->y_dtype == KTF_F32, "ktf_tdnn: F16 gemm writes half or fp32");
            KTF_REQUIRE(d->units > 128 && ldy % 4 == 0 && (d->act == KTF_ACT_RELU || d->act == KTF_ACT_NONE),
                        "ktf_tdnn: F16 gemm runs on the ring kernels only (units > 128, ldy %% 4 == 0, ReLU or no activation)");
        } else {
            KTF_REQUIRE(d->w_dtype == KTF_BF16, "ktf_tdnn: bf16 gemm needs bf16 weights");
            KTF_REQUIRE(d->y_dtype == KTF_BF16 || d->y_dtype == KTF_F32, "ktf_tdnn: bf16 gemm writes bf16 or fp32");
        }
        const bool x3 = d->gemm == KTF_GEMM_BF16X3;
        if (x3) KTF_REQUIRE((d->x_dtype == KTF_F32 || split_in) && w_lo, "ktf_tdnn: BF16X3 needs fp32 activations (or hi/lo planes) and w_lo");
        if (split_in) KTF_REQUIRE(d->units > 128 && ldy % 4 == 0, "ktf_tdnn_split: runs on the 256x256 kernel only (units > 128, ldy %% 4 == 0)");
        dim3 grid(ntiles, (unsigned)ktf_cdiv(Tout, BF_BM), (unsigned)B);
        // K-step: 64 when the per-context width allows it, else 32
        const bool k64 = (d->din_pad % 64) == 0;
#define BF_LAUNCH(BK, XF, X3)                                                                               \
    do {                                                                                                    \
        const size_t lds = (size_t)2 * 128 * BfCfg<BK>::PITCH * 2 * (X3 ? 2 : 1) * sizeof(unsigned short);  \
        if (lds > 64 * 1024)                                                                                \
            KTF_LDS_ONCE((int)lds, tdnn_bf16_kernel<BK, XF, X3>); \
        hipLaunchKernelGGL((tdnn_bf16_kernel<BK, XF, X3>), grid, dim3(256), lds, st, p);                     \
    } while (0)
        if (x3 && d->units > 128 && ldy % 4 == 0) {
            const int mtiles = ktf_cdiv(Tout, R_BM), ntiles_r = ktf_cdiv(d->units, R_BN);
            const int64_t gtiles = B * (int64_t)mtiles;
            const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles_r;
            KTF_REQUIRE(nblocks < (1ll << 31), "ktf_tdnn: grid too large");
#define X_LAUNCH1(A, ST, SP)                                                                                           \
    do {                                                                                                               \
        KTF_LDS_ONCE(X_LDS_BYTES, tdnn_x3r_kernel<A, ST, SP>); \
        hipLaunchKernelGGL((tdnn_x3r_kernel<A, ST, SP>), dim3((unsigned)nblocks), dim3(512), X_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
    } while (0)
#define X_LAUNCH(A)                                                                                                    \
    do {                                                                                                               \
        if (stats_sums) { if (split_in) X_LAUNCH1(A, true, true); else X_LAUNCH1(A, true, false); }                    \
        else { if (split_in) X_LAUNCH1(A, false, true); else X_LAUNCH1(A, false, false); }                            \
    } while (0)
#define XS_LAUNCH1(A, ST)                                                                                              \
    do {                                                                                                               \
        KTF_LDS_ONCE(XS_LDS_BYTES, tdnn_x3s_kernel<A, ST>); \
        hipLaunchKernelGGL((tdnn_x3s_kernel<A, ST>), dim3((unsigned)nblocks), dim3(512), XS_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
    } while (0)
#define XS_LAUNCH(A)                                                                                                   \
    do {                                                                                                               \
        if (stats_sums) XS_LAUNCH1(A, true); else XS_LAUNCH1(A, false);                                                \
    } while (0)
            const int x3s_env = KTF_KNOB("KTF_X3S", KTF_X3S_DEFAULT);   // probe builds: 0 = 32x32x16 kernels everywhere (A/B)
            // split planes in: the 16x16x32 kernel (x3s_env 2: also for the layers that write planes / fp32 out, else pooling only)
            if (split_in && (p.kinter || p.wtiled || p.xchunk || p.ychunk || (x3s_env && (stats_sums || x3s_env >= 2)))) {
                if (d->act == KTF_ACT_NONE) XS_LAUNCH(KTF_ACT_NONE);
                else if (d->act == KTF_ACT_RELU) XS_LAUNCH(KTF_ACT_RELU);
                else if (d->act == KTF_ACT_SIGMOID) XS_LAUNCH(KTF_ACT_SIGMOID);
                else XS_LAUNCH(KTF_ACT_TANH);
            } else
            if (d->act == KTF_ACT_NONE) X_LAUNCH(KTF_ACT_NONE);
            else if (d->act == KTF_ACT_RELU) X_LAUNCH(KTF_ACT_RELU);
            else if (d->act == KTF_ACT_SIGMOID) X_LAUNCH(KTF_ACT_SIGMOID);
            else X_LAUNCH(KTF_ACT_TANH);
#undef XS_LAUNCH
#undef XS_LAUNCH1
#undef X_LAUNCH
#undef X_LAUNCH1
        } else if (x3) {
            if (k64) BF_LAUNCH(64, true, true); else BF_LAUNCH(32, true, true);
        } else if (d->x_dtype == KTF_F32) {
            if (k64) BF_LAUNCH(64, true, false); else BF_LAUNCH(32, true, false);
        } else {
            KTF_REQUIRE(d->x_dtype == (f16 ? KTF_F16 : KTF_BF16), "ktf_tdnn: bad x_dtype");
            if (d->units > 128 && ldy % 4 == 0) {
                // W must be padded to a multiple of 256 rows for this kernel (documented in ktf_hip.h)
                const int mtiles = ktf_cdiv(Tout, R_BM), ntiles_r = ktf_cdiv(d->units, R_BN);
                const int64_t gtiles = B * (int64_t)mtiles;
                const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles_r;
                KTF_REQUIRE(nblocks < (1ll << 31), "ktf_tdnn: grid too large");
#define R_LAUNCH(A)                                                                                                    \
    do {                                                                                                               \
        if (stats_sums) {                                                                                              \
            KTF_LDS_ONCE(R_LDS_BYTES, tdnn_bf16r_kernel<A, true>); \
            hipLaunchKernelGGL((tdnn_bf16r_kernel<A, true>), dim3((unsigned)nblocks), dim3(512), R_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(R_LDS_BYTES, tdnn_bf16r_kernel<A, false>); \
            hipLaunchKernelGGL((tdnn_bf16r_kernel<A, false>), dim3((unsigned)nblocks), dim3(512), R_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, (double*)nullptr); \
        }                                                                                                              \
    } while (0)
                // 128x256 tiles with two workgroups per CU win while the fixed per-tile phases are comparable to the K-loop
                // (K <= 768); deeper K amortises them and the 256x256 tile moves fewer bytes per flop (probe builds: KTF_HTILE=0/1 forces).
                const int htile_env = KTF_KNOB("KTF_HTILE", -1);
                const bool htile = htile_env >= 0 ? (htile_env != 0) : (p.ktot <= 768);
                if (htile && (d->act == KTF_ACT_RELU || d->act == KTF_ACT_NONE)) {
                    const int mt_h = ktf_cdiv(Tout, H_BM);
                    const int64_t gt_h = B * (int64_t)mt_h;
                    const int64_t nb_h = ((gt_h + 7) / 8) * 8 * ntiles_r;
                    KTF_REQUIRE(nb_h < (1ll << 31), "ktf_tdnn: grid too large");
                    long long* const dbgptr = KTF_PROBE_BUF;
#define H_LAUNCH(A, ST)                                                                                                \
    do {                                                                                                               \
        if (f16) {                                                                                                     \
            KTF_LDS_ONCE(H_LDS_BYTES, tdnn_bf16h_kernel<A, ST, true>); \
            hipLaunchKernelGGL((tdnn_bf16h_kernel<A, ST, true>), dim3((unsigned)nb_h), dim3(256), H_LDS_BYTES, st, p, mt_h, ntiles_r, (int)gt_h, stats_sums, dbgptr); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(H_LDS_BYTES, tdnn_bf16h_kernel<A, ST, false>); \
            hipLaunchKernelGGL((tdnn_bf16h_kernel<A, ST, false>), dim3((unsigned)nb_h), dim3(256), H_LDS_BYTES, st, p, mt_h, ntiles_r, (int)gt_h, stats_sums, dbgptr); \
        }                                                                                                              \
    } while (0)
                    if (d->act == KTF_ACT_RELU) { if (stats_sums) H_LAUNCH(KTF_ACT_RELU, true); else H_LAUNCH(KTF_ACT_RELU, false); }
                    else { if (stats_sums) H_LAUNCH(KTF_ACT_NONE, true); else H_LAUNCH(KTF_ACT_NONE, false); }
#undef H_LAUNCH
                    KTF_CHECK_LAUNCH("ktf_tdnn");
                    return KTF_OK;
                }
                const int mfma16 = KTF_KNOB("KTF_MFMA16", 1);     // 16x16x32 variant (probe builds: 0 = 32x32x16, A/B)
                if ((mfma16 || f16) && (d->act == KTF_ACT_RELU || d->act == KTF_ACT_NONE)) {
#define S_LAUNCH(A, ST)                                                                                                \
    do {                                                                                                               \
        if (f16) {                                                                                                     \
            KTF_LDS_ONCE(R16_LDS_BYTES, tdnn_bf16r16_kernel<A, ST, true>); \
            hipLaunchKernelGGL((tdnn_bf16r16_kernel<A, ST, true>), dim3((unsigned)nblocks), dim3(512), R16_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(R16_LDS_BYTES, tdnn_bf16r16_kernel<A, ST, false>); \
            hipLaunchKernelGGL((tdnn_bf16r16_kernel<A, ST, false>), dim3((unsigned)nblocks), dim3(512), R16_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        }                                                                                                              \
    } while (0)
                    if (d->act == KTF_ACT_RELU) { if (stats_sums) S_LAUNCH(KTF_ACT_RELU, true); else S_LAUNCH(KTF_ACT_RELU, false); }
                    else { if (stats_sums) S_LAUNCH(KTF_ACT_NONE, true); else S_LAUNCH(KTF_ACT_NONE, false); }
#undef S_LAUNCH
                    KTF_CHECK_LAUNCH("ktf_tdnn");
                    return KTF_OK;
                }
                if (d->act == KTF_ACT_NONE) R_LAUNCH(KTF_ACT_NONE);
                else if (d->act == KTF_ACT_RELU) R_LAUNCH(KTF_ACT_RELU);
                else if (d->act == KTF_ACT_SIGMOID) R_LAUNCH(KTF_ACT_SIGMOID);
                else R_LAUNCH(KTF_ACT_TANH);
#undef R_LAUNCH
            } else if (k64 && ldy % 4 == 0) {
                const int mtiles = ktf_cdiv(Tout, G_BM), ntiles_g = ktf_cdiv(d->units, G_BN);
                const int64_t gtiles = B * (int64_t)mtiles;
                const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles_g;
                KTF_REQUIRE(nblocks < (1ll << 31), "ktf_tdnn: grid too large");
                KTF_LDS_ONCE(G_LDS_BYTES, tdnn_bf16g_kernel);
                hipLaunchKernelGGL(tdnn_bf16g_kernel, dim3((unsigned)nblocks), dim3(256), G_LDS_BYTES, st, p, mtiles, ntiles_g, (int)gtiles);
            } else if (k64) BF_LAUNCH(64, false, false); else BF_LAUNCH(32, false, false);
        }
#undef BF_LAUNCH
    } else {
        KTF_REQUIRE(false, "ktf_tdnn: unknown gemm mode %d", d->gemm);
    }
    KTF_CHECK_LAUNCH("ktf_tdnn");
    return KTF_OK;
}

extern "C" int ktf_tdnn(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
                        const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift,
                        void* y, int64_t ldy, int32_t* out_lens, void* stream) {
    KTF_REQUIRE(y, "ktf_tdnn: null output");
    return tdnn_launch(x, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, y, ldy, out_lens, nullptr, stream);
}

extern "C" int ktf_tdnn_stats(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
                              const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift,
                              double* sums, void* stream) {
    KTF_REQUIRE(sums, "ktf_tdnn_stats: null sums");
    return tdnn_launch(x, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, nullptr, 0, nullptr, sums, stream);
}

extern "C" int ktf_tdnn_split(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* lens,
                              const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                              const float* shift, void* y, void* y_lo, int64_t ldy, int32_t* out_lens, void* stream) {
    KTF_REQUIRE(y && d, "ktf_tdnn_split: null argument");
    KTF_REQUIRE((d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16) || d->gemm == KTF_GEMM_F16X2,
                "ktf_tdnn_split: needs KTF_GEMM_BF16X3 with x_dtype KTF_BF16 (hi/lo planes) or KTF_GEMM_F16X2 (one half plane)");
    return tdnn_launch(x_hi, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, y, ldy, out_lens, nullptr, stream, x_lo, y_lo);
}

extern "C" int ktf_tdnn_split_stats(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx,
                                    const int32_t* lens, const KtfTdnnDesc* d, const void* w, const void* w_lo,
                                    const float* bias, const float* scale, const float* shift, double* sums, void* stream) {
    KTF_REQUIRE(sums && d, "ktf_tdnn_split_stats: null argument");
    KTF_REQUIRE((d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16) || d->gemm == KTF_GEMM_F16X2,
                "ktf_tdnn_split_stats: needs KTF_GEMM_BF16X3 with x_dtype KTF_BF16 (hi/lo planes) or KTF_GEMM_F16X2 (one half plane)");
    return tdnn_launch(x_hi, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, nullptr, 0, nullptr, sums, stream, x_lo, nullptr);
}

// fp32 rows -> the two bf16 planes of the split representation (hi = bf16(v), lo = bf16(v - hi)); pad columns zero
__global__ void split_bf16_kernel(const float* __restrict__ src, int64_t rows, int D, int64_t lds_, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo, int64_t ldd) {
    const int64_t total = rows * ldd;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / ldd;
        const int d = (int)(e - r * ldd);
        const float v = d < D ? src[r * lds_ + d] : 0.0f;
        const unsigned short h = f2bf(v);
        hi[e] = h;
        lo[e] = f2bf(v - bf2f(h));
    }
}

extern "C" int ktf_split_bf16(const float* src, int64_t rows, int32_t D, int64_t ld_src, void* hi, void* lo, int64_t ld_dst,
                              void* stream) {
    KTF_REQUIRE(src && hi && lo, "ktf_split_bf16: null argument");
    KTF_REQUIRE(rows >= 0 && D > 0 && ld_src >= D && ld_dst >= D, "ktf_split_bf16: bad sizes");
    if (rows == 0) return KTF_OK;
    const int64_t total = rows * ld_dst;
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, rows, D, ld_src,
                       (unsigned short*)hi, (unsigned short*)lo, ld_dst);
    KTF_CHECK_LAUNCH("ktf_split_bf16");
    return KTF_OK;
}

// 128-row blocks, rounded up to whole 256-row tiles (a 256-row tile always writes both of its blocks)
extern "C" int64_t ktf_stats_slots(int64_t T) { return T <= 0 ? 2 : 2 * ((T + 255) / 256); }

// mean / std from the fp64 column sums of ktf_tdnn_stats: out[b, c] = mean, out[b, D + c] = sqrt(max(E[x^2]-mean^2,0)+eps)
__global__ void stats_finalize_kernel(const double* __restrict__ sums, int64_t slots, const int32_t* __restrict__ lens, int64_t T,
                                      int64_t B, int D, int include_std, float eps, float* __restrict__ out, int64_t ldo) {
    const int64_t total = B * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / D;
        const int c = (int)(e - b * D);
        const int len = lens ? lens[b] : (int)T;
        const double n = (double)len;
        double s = 0.0, q = 0.0;
        if (slots == 0) {
            s = sums[(b * 2) * D + c];
            q = sums[(b * 2 + 1) * D + c];
        } else {
            const int used = (len + 127) >> 7;                 // blocks holding valid rows, added in block order
            for (int k = 0; k < used; ++k) {
                s += sums[((b * slots + k) * 2) * D + c];
                q += sums[((b * slots + k) * 2 + 1) * D + c];
            }
        }
        const double mean = s / n;
        out[b * ldo + c] = (float)mean;
        if (include_std) {
            const double var = q / n - mean * mean;
            out[b * ldo + D + c] = (float)sqrt(fmax(var, 0.0) + (double)eps);
        }
    }
}

extern "C" int ktf_stats_finalize(const double* sums, const int32_t* lens, int64_t T, int64_t B, int32_t D,
                                  int32_t include_std, float eps, float* out, int64_t ld_out, void* stream) {
    KTF_REQUIRE(sums && out, "ktf_stats_finalize: null argument");
    KTF_REQUIRE(B >= 0 && D > 0 && ld_out >= (include_std ? 2 : 1) * (int64_t)D, "ktf_stats_finalize: bad sizes");
    if (B == 0) return KTF_OK;
    int blocks = ktf_cdiv(B * D, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sums, (int64_t)0, lens, T, B, D, include_std, eps, out, ld_out);
    KTF_CHECK_LAUNCH("ktf_stats_finalize");
    return KTF_OK;
}

extern "C" int ktf_stats_finalize_slots(const double* sums, int64_t slots, const int32_t* lens, int64_t T, int64_t B, int32_t D,
                                        int32_t include_std, float eps, float* out, int64_t ld_out, void* stream) {
    KTF_REQUIRE(sums && out, "ktf_stats_finalize_slots: null argument");
    KTF_REQUIRE(B >= 0 && D > 0 && ld_out >= (include_std ? 2 : 1) * (int64_t)D, "ktf_stats_finalize_slots: bad sizes");
    KTF_REQUIRE(slots >= ktf_stats_slots(T), "ktf_stats_finalize_slots: %lld slots < ktf_stats_slots(%lld)", (long long)slots, (long long)T);
    if (B == 0) return KTF_OK;
    int blocks = ktf_cdiv(B * D, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sums, slots, lens, T, B, D, include_std, eps, out, ld_out);
    KTF_CHECK_LAUNCH("ktf_stats_finalize_slots");
    return KTF_OK;
}

extern "C" int ktf_affine_act_f32(const float* x, int64_t rows, int32_t D, int32_t act, const float* scale,
                                  const float* shift, float* y, void* stream) {
    KTF_REQUIRE(x && y, "ktf_affine_act_f32: null argument");
    KTF_REQUIRE(rows >= 0 && D > 0, "ktf_affine_act_f32: bad sizes");
    KTF_REQUIRE(act >= KTF_ACT_NONE && act <= KTF_ACT_TANH, "ktf_affine_act_f32: bad activation");
    const int64_t total = rows * D;
    if (total == 0) return KTF_OK;
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(affine_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, total, D, act, scale, shift, y);
    KTF_CHECK_LAUNCH("ktf_affine_act_f32");
    return KTF_OK;
}

extern "C" int ktf_convert_pad(const void* src, int32_t src_dtype, int64_t rows, int32_t D, int64_t ld_src, void* dst,
                               int32_t dst_dtype, int64_t ld_dst, void* stream) {
    KTF_REQUIRE(src && dst, "ktf_convert_pad: null argument");
    KTF_REQUIRE(rows >= 0 && D > 0 && ld_src >= D && ld_dst >= D, "ktf_convert_pad: bad sizes");
    const int64_t total = rows * ld_dst;
    if (total == 0) return KTF_OK;
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
#define CP_CASE(SD, ST, DD, DT)                                                                                        \
    if (src_dtype == SD && dst_dtype == DD) {                                                                          \
        hipLaunchKernelGGL((convert_pad_kernel<ST, DT>), dim3(blocks), dim3(256), 0, st, (const ST*)src, rows, D, ld_src, \
                           (DT*)dst, ld_dst);                                                                          \
        launched = true;                                                                                               \
    }
    bool launched = false;
    CP_CASE(KTF_F32, float, KTF_F32, float) CP_CASE(KTF_F32, float, KTF_BF16, unsigned short) CP_CASE(KTF_F32, float, KTF_F16, _Float16)
    CP_CASE(KTF_BF16, unsigned short, KTF_F32, float) CP_CASE(KTF_BF16, unsigned short, KTF_BF16, unsigned short)
    CP_CASE(KTF_F16, _Float16, KTF_F32, float) CP_CASE(KTF_F16, _Float16, KTF_F16, _Float16)
#undef CP_CASE
    KTF_REQUIRE(launched, "ktf_convert_pad: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
    KTF_CHECK_LAUNCH("ktf_convert_pad");
    return KTF_OK;
}
