// KTF_GEMM_BF16 (one bf16 MFMA pass; outside the 1e-4 tolerance, BASELINE config 3's precision) and the small-layer
// forms of KTF_GEMM_BF16X3: 128 x 128 register-/DMA-staged tiles, the 256 x 256 ring kernels on 32x32x16 (sigmoid / tanh) and
// 16x16x32 MFMAs, and the 128 x 256 two-workgroups-per-CU kernel for K <= 768.
#include "tdnn_ring.h"

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
template <int ACT>
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
            pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(stg + i * 16 * R16_PK_PITCH + j * 32) = pk;
        }
    }
    __syncthreads();
    r16_store_staged(p, rsm, reinterpret_cast<unsigned short*>(p.y), b, t0, n0, out_len, wave, lane);
}

template <int ACT, bool STATS>
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
            for (int j = 0; j < 4; ++j) acc[i][j] = STATS ? mfma16x16x32(a[i], bq[j], acc[i][j]) : mfma16x16x32(bq[j], a[i], acc[i][j]);
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
            for (int j = 0; j < 4; ++j) acc[i][j] = STATS ? mfma16x16x32(a[i], bq[j], acc[i][j]) : mfma16x16x32(bq[j], a[i], acc[i][j]);
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
        ring_epilogue16_pk<ACT>(acc, p, rsm, b, t0, n0, out_len, wm, wn, wave, lane);
    }
}

template <int ACT, bool STATS>
__global__ __launch_bounds__(512) void tdnn_bf16r16_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
                                                           double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    r16_tile<ACT, STATS>(p, mtiles, ntiles, gtiles, stats, rsm, blockIdx.x);
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


template <int ACT, bool STATS>
__global__ __launch_bounds__(256, 2) void tdnn_bf16h_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
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
        const unsigned vo_ = (unsigned)r_ * ldxb + a_cb[i] + (unsigned)is_db;                                          \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + vo_),                                                       \
            (lds_ptr_t*)(rsm + is_slot * H_STAGE_BYTES + wn * 1024 + (i) * 4096), 16, 0, KTF_AUX_A);                   \
    }
#define H_DMA_B(i)                                                                                                     \
    {                                                                                                                  \
        const unsigned vo_ = w_ob[i] + (unsigned)(is_ks * (R_BK * 2));                                                 \
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
        if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool refill = is_ks < nk;
        const unsigned char* sa = rsm + cs * H_STAGE_BYTES;
        const unsigned char* sb = sa + H_A_BYTES;
        cs = (cs == H_NSTAGE - 1) ? 0 : cs + 1;
        {
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
                acc[i][j] = STATS ? mfma16x16x32(a[i], bq[j], acc[i][j]) : mfma16x16x32(bq[j], a[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {
#pragma unroll
                for (int i2 = 4; i2 < 8; ++i2) a[i2] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + i2 * 16 * 64);
            }
            if (refill) {
                if (i == 1) H_DMA_A(0)
                if (i == 2) H_DMA_A(1)
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
                pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(stg + i * 16 * H_PK_PITCH + j * 32) = pk;
            }
        }
    }
    __syncthreads();
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
}


// ------------------------------------------------------------------------------------ launcher
int tdnn_launch_16(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, int64_t ldy, double* stats_sums, hipStream_t st) {
    const unsigned ntiles = (unsigned)ktf_cdiv(d->units, 128);
    {
        KTF_REQUIRE(d->w_dtype == KTF_BF16, "ktf_tdnn: bf16 gemm needs bf16 weights");
        KTF_REQUIRE(d->y_dtype == KTF_BF16 || d->y_dtype == KTF_F32, "ktf_tdnn: bf16 gemm writes bf16 or fp32");
        const bool x3 = d->gemm == KTF_GEMM_BF16X3;
        dim3 grid(ntiles, (unsigned)ktf_cdiv(Tout, BF_BM), (unsigned)B);
        // K-step: 64 when the per-context width allows it, else 32
        const bool k64 = (d->din_pad % 64) == 0;
#define BF_LAUNCH(BK, XF, X3)                                                                               \
    do {                                                                                                    \
        const size_t lds = (size_t)2 * 128 * BfCfg<BK>::PITCH * 2 * (X3 ? 2 : 1) * sizeof(unsigned short);  \
        KTF_NOTE_KERNEL("tdnn_bf16_kernel<" #BK ", " #XF ", " #X3 ">");                                      \
        if (lds > 64 * 1024)                                                                                \
            KTF_LDS_ONCE((int)lds, tdnn_bf16_kernel<BK, XF, X3>); \
        hipLaunchKernelGGL((tdnn_bf16_kernel<BK, XF, X3>), grid, dim3(256), lds, st, p);                     \
    } while (0)
        if (x3) {
            if (k64) BF_LAUNCH(64, true, true); else BF_LAUNCH(32, true, true);
        } else if (d->x_dtype == KTF_F32) {
            if (k64) BF_LAUNCH(64, true, false); else BF_LAUNCH(32, true, false);
        } else {
            KTF_REQUIRE(d->x_dtype == KTF_BF16, "ktf_tdnn: bad x_dtype");
            if (d->units > 128 && ldy % 4 == 0) {
                // W must be padded to a multiple of 256 rows for this kernel (documented in ktf_hip.h)
                const int mtiles = ktf_cdiv(Tout, R_BM), ntiles_r = ktf_cdiv(d->units, R_BN);
                const int64_t gtiles = B * (int64_t)mtiles;
                const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles_r;
                KTF_REQUIRE(nblocks < (1ll << 31), "ktf_tdnn: grid too large");
#define R_LAUNCH(A)                                                                                                    \
    do {                                                                                                               \
        KTF_NOTE_KERNEL("tdnn_bf16r_kernel");                                                                          \
        if (stats_sums) {                                                                                              \
            KTF_LDS_ONCE(R_LDS_BYTES, tdnn_bf16r_kernel<A, true>); \
            hipLaunchKernelGGL((tdnn_bf16r_kernel<A, true>), dim3((unsigned)nblocks), dim3(512), R_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(R_LDS_BYTES, tdnn_bf16r_kernel<A, false>); \
            hipLaunchKernelGGL((tdnn_bf16r_kernel<A, false>), dim3((unsigned)nblocks), dim3(512), R_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, (double*)nullptr); \
        }                                                                                                              \
    } while (0)
                // 128x256 tiles with two workgroups per CU win while the fixed per-tile phases are comparable to the K-loop
                // (K <= 768); deeper K amortises them and the 256x256 tile moves fewer bytes per flop
                const bool htile = p.ktot <= 768;
                if (htile && (d->act == KTF_ACT_RELU || d->act == KTF_ACT_NONE)) {
                    const int mt_h = ktf_cdiv(Tout, H_BM);
                    const int64_t gt_h = B * (int64_t)mt_h;
                    const int64_t nb_h = ((gt_h + 7) / 8) * 8 * ntiles_r;
                    KTF_REQUIRE(nb_h < (1ll << 31), "ktf_tdnn: grid too large");
#define H_LAUNCH(A, ST)                                                                                                \
    do {                                                                                                               \
        KTF_NOTE_KERNEL("tdnn_bf16h_kernel");                                                                          \
        KTF_LDS_ONCE(H_LDS_BYTES, tdnn_bf16h_kernel<A, ST>);                                                           \
        hipLaunchKernelGGL((tdnn_bf16h_kernel<A, ST>), dim3((unsigned)nb_h), dim3(256), H_LDS_BYTES, st, p, mt_h, ntiles_r, (int)gt_h, stats_sums); \
    } while (0)
                    if (d->act == KTF_ACT_RELU) { if (stats_sums) H_LAUNCH(KTF_ACT_RELU, true); else H_LAUNCH(KTF_ACT_RELU, false); }
                    else { if (stats_sums) H_LAUNCH(KTF_ACT_NONE, true); else H_LAUNCH(KTF_ACT_NONE, false); }
#undef H_LAUNCH
                    KTF_CHECK_LAUNCH("ktf_tdnn");
                    return KTF_OK;
                }
                if (d->act == KTF_ACT_RELU || d->act == KTF_ACT_NONE) {       // 16x16x32 MFMAs; sigmoid / tanh stay on the 32x32x16 kernel
#define S_LAUNCH(A, ST)                                                                                                \
    do {                                                                                                               \
        KTF_NOTE_KERNEL("tdnn_bf16r16_kernel");                                                                        \
        KTF_LDS_ONCE(R16_LDS_BYTES, tdnn_bf16r16_kernel<A, ST>);                                                       \
        hipLaunchKernelGGL((tdnn_bf16r16_kernel<A, ST>), dim3((unsigned)nblocks), dim3(512), R16_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
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
                KTF_NOTE_KERNEL("tdnn_bf16g_kernel");
                KTF_LDS_ONCE(G_LDS_BYTES, tdnn_bf16g_kernel);
                hipLaunchKernelGGL(tdnn_bf16g_kernel, dim3((unsigned)nblocks), dim3(256), G_LDS_BYTES, st, p, mtiles, ntiles_g, (int)gtiles);
            } else if (k64) BF_LAUNCH(64, false, false); else BF_LAUNCH(32, false, false);
        }
#undef BF_LAUNCH
    }
    KTF_CHECK_LAUNCH("ktf_tdnn");
    return KTF_OK;
}
