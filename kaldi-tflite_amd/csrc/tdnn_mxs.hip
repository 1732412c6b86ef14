// KTF_GEMM_F16MX for MULTI-CONTEXT layers on activation SLABS (the 256 x 256 eight-wave tile of tdnn_mx.hip, same planes, same weight
// images, same arithmetic, same epilogue -- only what the K-loop fetches differs).
//
// The K-steps of a layer walk (32-feature chunk, context offset): the three (five) K-steps of one chunk read THE SAME rows of that
// chunk's planes, shifted by the context offsets. tdnn_mx.hip gathers them three times (a 16 KiB half stage + 9 KiB of side data per
// K-step). Here a chunk's rows [t0 - 4, t0 + 268) are fetched ONCE as a slab -- 17 KiB of half values + 9.6 KiB of e2m1 codes and scale
// words -- and every K-step of the chunk reads its fragments from the slab at its own row shift: the fragment address takes the offset,
// the DMA does not. For tdnn2 / tdnn3 (three contexts) a super-step then issues ~150 LDS-DMA instructions per CU instead of 224, and
// the K-loop's time follows the instruction count (a super-step costs its MFMA-only time plus ~9 ns per vector-
// memory instruction the CU issues).
//
// The slab holds clamped rows: position s = row clamp(t0 - 4 + s, 0, len - 1) of the utterance, so a fragment read at shift `off` sees
// clamp(t + off) -- SAME padding = edge replication (layers/tdnn/tdnn.py:246-247) -- because M-tiles never straddle utterances here.
// Requirements: 2 <= nctx, |context offset| <= 4 (mx_launch checks; everything else runs tdnn_mx_kernel).
//
// LDS (161,792 B): half slabs 2 x 17 KiB (chunk parity) | W half stages 2 x 16 KiB (K-step parity) | side slabs 4 x 11,520 B (chunk & 3:
// the four K blocks of a super-step touch at most three chunks) | side W 44 KiB | epilogue constants 3 KiB.
//
// Replaces: layers/tdnn/tdnn.py:251-280 (+ keras ReLU, batchnorm.py:78-88, stats_pooling.py:211-240 when fused).
#include "tdnn_mx_common.h"

#define XS_PAD 4
#define XS_ROWS 272                                  // 256 + 2 * XS_PAD, in whole 16-row DMA pieces
#define XS_ASLAB (XS_ROWS * 64)                      // 17,408 B: one chunk's half values, 64-byte rows with the XOR piece placement
#define XS_W_OFF (2 * XS_ASLAB)
#define XS_SS_OFF (XS_W_OFF + 2 * MX_TILE)
#define XS_SROWS 320                                 // rows of a side plane: five whole 64-row DMA pieces (272 are read)
#define XS_SPLANE (XS_SROWS * 16)                    // 5,120 B: one e2m1 plane of a side slab
#define XS_SSLAB (2 * XS_SPLANE + XS_SROWS * 4)      // 11,520 B: xl4 | x4 | scale words
#define XS_SW_OFF (XS_SS_OFF + 4 * XS_SSLAB)
#define XS_SW_BYTES 45056                            // the used part of a (N-tile, super-step) block of wq
#define XS_PRM_OFF (XS_SW_OFF + XS_SW_BYTES)
#define XS_LDS_BYTES (XS_PRM_OFF + 3 * 256 * 4)      // 161,792 B

static_assert(XS_LDS_BYTES <= 163840 && 8 * 64 * MX_EPW_PITCH * 4 <= XS_PRM_OFF, "LDS budget (K-loop, epilogue staging image)");

template <int ACT, int OUT>
__device__ __forceinline__ void mxs_tile(const MxParams& p, const int id, int mtiles, int ntiles, int gtiles, double* __restrict__ stats,
                                         unsigned char* rsm) {
    const int xcd = id & 7, slot = id >> 3;             // an XCD runs all N-tiles of an M-tile back to back (its L2 keeps the A rows)
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = g / mtiles, mt = g - b * mtiles;
    const int n0 = nt * 256, t0 = mt * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int len = p.lens ? p.lens[b] : (int)p.T;
    if (t0 >= len || len <= 0) return;
    const int lenm1 = len - 1, out_len = len;            // (SAME padding, no subsampling: mx_launch)
    const unsigned Tu = (unsigned)p.T;
    const int64_t ub = (int64_t)b * p.nch_in * p.T;       // first (chunk, row) record of this utterance
    const char* xh = p.xh + ub * 64;
    const char* xl4 = p.xl4 + ub * 16;
    const char* x4 = p.x4 + ub * 16;
    const char* xs = p.xs + ub * 4;
    const int nkp = p.nss * 4;
    const char* wh = p.wh + (int64_t)nt * nkp * MX_TILE;
    const char* wq = p.wq + (int64_t)nt * p.nss * MX_WQ_BLOCK;
    const unsigned long long cpk0 = p.ctx_pk[0], cpk1 = p.ctx_pk[1];
#define XS_CTX(ci_) ((int)(signed char)(((ci_) < 8 ? cpk0 : cpk1) >> (((ci_) & 7) * 8)))

    // ---- slab DMAs. A chunk's slab is 32 pieces of work, four ("slots") per wave: piece q = 8 s + wave (s = 0..3):
    //   q = 0..16   half values: rows 16 q .. 16 q + 15 of the slab, 64 B each (lane >> 2 = row, lane & 3 = 16-byte position, which holds
    //               chunk (lane & 3) ^ f(row): conflict-free 16-byte fragment reads at ANY row shift -- 16 consecutive rows always hold
    //               every (row & 3, (row >> 2) & 3) pair once)
    //   q = 17..31  side pieces sp = q - 17: kind sp / 5 (e2m1 codes of the residual, of the value, scale words), rows 64 (sp % 5) + lane
    //               (the planes hold 320 rows so that every piece is whole; rows past 271 are never read)
    // Everything but the chunk is fixed per (wave, slot, lane): a per-lane byte offset inside the chunk's plane block (the clamped
    // row), a uniform plane pointer, record size and LDS position.
    unsigned sl_vo[4];
    const char* sl_ptr[4];
    unsigned sl_cs[4];                                // chunk stride of the slot's plane: T * record size
    int sl_lds[4];                                    // LDS offset inside the half slab / the side slab
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int q = s * 8 + wave;
        if (q < 17) {
            int r = t0 - XS_PAD + q * 16 + (lane >> 2);
            r = r < 0 ? 0 : (r > lenm1 ? lenm1 : r);
            const int hrow = lane >> 2;
            sl_vo[s] = (unsigned)r * 64u + (unsigned)(((lane & 3) ^ ((4 - ((hrow >> 2) & 3)) & 3)) * 16);
            sl_ptr[s] = xh;
            sl_cs[s] = Tu * 64u;
            sl_lds[s] = q * 1024;
        } else {
            const int sp = q - 17, kind = sp / 5, ps = sp - kind * 5;
            int r = t0 - XS_PAD + ps * 64 + lane;
            r = r < 0 ? 0 : (r > lenm1 ? lenm1 : r);
            const unsigned rs = kind == 2 ? 4u : 16u;
            sl_vo[s] = (unsigned)r * rs;
            sl_ptr[s] = kind == 0 ? xl4 : (kind == 1 ? x4 : xs);
            sl_cs[s] = Tu * rs;
            sl_lds[s] = kind * XS_SPLANE + ps * (kind == 2 ? 256 : 1024);
        }
    }
    const bool s2_half = wave == 0;                   // slot 2: piece 16 of the half slab on wave 0, side pieces elsewhere
    const bool s3_words = wave >= 3;                  // slot 3: 4-byte scale words on waves 3..7
#define XS_SLAB_DMA(c_, ptr_, cs_, vo_, half_, lds_, words_)                                                             \
    {                                                                                                                  \
        const char* g_ = (ptr_) + (unsigned)(c_) * (cs_);                                                              \
        unsigned v_ = (vo_);                                                                                           \
        asm volatile("" : "+s"(g_), "+v"(v_));       /* (a scalar base + a 32-bit lane offset, not a 64-bit lane address) */ \
        unsigned char* l_ = rsm + ((half_) ? ((c_) & 1) * XS_ASLAB : XS_SS_OFF + ((c_) & 3) * XS_SSLAB) + (lds_);      \
        if (words_) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(g_ + v_), (lds_ptr_t*)l_, 4, 0, 0);                  \
        else __builtin_amdgcn_global_load_lds((glb_ptr_t*)(g_ + v_), (lds_ptr_t*)l_, 16, 0, 0);                        \
    }
#define XS_SLAB(c_, s_) XS_SLAB_DMA(c_, sl_ptr[s_], sl_cs[s_], sl_vo[s_], (s_) < 2 || ((s_) == 2 && s2_half), sl_lds[s_], (s_) == 3 && s3_words)
    // slot 1 (first_) or slot 2: one instruction with selected operands (the two are never due in the same K-step)
#define XS_SLAB_1OR2(c_, first_)                                                                                       \
    XS_SLAB_DMA(c_, (first_) ? sl_ptr[1] : sl_ptr[2], (first_) ? sl_cs[1] : sl_cs[2], (first_) ? sl_vo[1] : sl_vo[2], (first_) || s2_half, \
                (first_) ? sl_lds[1] : sl_lds[2], false)
    // W half stage of K-step ks_: piece n_ = 0, 1 (8 KiB each)
#define XS_DMA_W(ks_, n_)                                                                                              \
    {                                                                                                                  \
        const unsigned vo_ = (unsigned)(ks_) * (unsigned)MX_TILE + (unsigned)((n_) * 512 + tid) * 16u;                 \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wh + vo_), (lds_ptr_t*)(rsm + XS_W_OFF + ((ks_) & 1) * MX_TILE + (n_) * 8192 + wave * 1024), 16, 0, 0); \
    }
    // side W of super-step ss_: piece n_ = 0..5 of the block's 44 KiB
#define XS_DMA_SW(ss_, n_)                                                                                             \
    {                                                                                                                  \
        const int idx_ = (n_) * 8 + wave;                                                                              \
        if (idx_ < 44) {                                                                                               \
            const unsigned vo_ = (unsigned)(ss_) * (unsigned)MX_WQ_BLOCK + (unsigned)idx_ * 1024u + (unsigned)lane * 16u; \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wq + vo_), (lds_ptr_t*)(rsm + XS_SW_OFF + idx_ * 1024), 16, 0, 0); \
        }                                                                                                              \
    }
    // ---- prologue: the slab of chunk 0 and the W stage of K-step 0
    XS_DMA_W(0, 0) XS_DMA_W(0, 1)
    XS_SLAB(0, 0) XS_SLAB(0, 1) XS_SLAB(0, 2) XS_SLAB(0, 3)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    if (tid < 256) {                                 // epilogue constants of the tile's columns (read behind the K-loop)
        float* prm = reinterpret_cast<float*>(rsm + XS_PRM_OFF);
        const int n = n0 + tid;
        const bool nv = n < p.units;
        prm[tid] = (nv && p.bias) ? p.bias[n] : 0.0f;
        prm[256 + tid] = (nv && p.scale) ? p.scale[n] : 1.0f;
        prm[512 + tid] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }

    const int r16 = lane & 15, q4 = lane >> 4;
    const int frb = (4 - ((r16 >> 2) & 3)) & 3;
    const int b_row_off = (wn * 64 + r16) * 64 + ((q4 ^ frb) << 4);
    const int sw_col = q4 * 256 + wn * 64 + r16;           // side W record of column block 0
    const int arow0 = wm * 128 + r16 + XS_PAD;             // slab row of this lane's row of row block 0 at offset 0

    // the K-step being computed: chunk, context index (scalars, advanced once per K-step)
    int k_c = 0, k_ci = 0;
    const int s2_ci = 2 % p.nctx, s3_ci = 3 % p.nctx;
    // this lane's K block of the super-step being computed (lane quarter q4 = K block): chunk and context index of K-step 4 ss + q4
    int m_ci = q4 % p.nctx, m_c = q4 / p.nctx;

    for (int ss = 0; ss < p.nss; ++ss) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ks = 4 * ss + j;
            // everything issued during the previous K-step has landed, except (j == 2) the side-W pieces issued last in it
            if (j == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool next = ks + 1 < nkp;
            const bool live = ks < p.nk;              // (padded K-steps: zero weights; they read the current chunk's slab)
            const int off = XS_CTX(k_ci);
            const unsigned char* sa = rsm + (k_c & 1) * XS_ASLAB;
            const unsigned char* sw = rsm + XS_W_OFF + (ks & 1) * MX_TILE;
            const int rowb = arow0 + off;
            const int a_row_off = rowb * 64 + ((q4 ^ ((4 - ((rowb >> 2) & 3)) & 3)) << 4);
            hfrag8 bh[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) bh[jj] = *reinterpret_cast<const hfrag8*>(sw + b_row_off + jj * 1024);
            hfrag8 a_cur = *reinterpret_cast<const hfrag8*>(sa + a_row_off);
            // the K-step's DMAs go out one or two at a time between the row blocks' MFMAs: the next W stage, this K-step's share of the
            // NEXT chunk's slab (slot s at context index s % nctx), and in F1 the W side of this super-step
            const bool slab_next = live && k_c + 1 < p.nch_in;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                hfrag8 a_nxt = a_cur;
                if (i < 7) a_nxt = *reinterpret_cast<const hfrag8*>(sa + a_row_off + (i + 1) * 1024);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);
                a_cur = a_nxt;
                __builtin_amdgcn_sched_barrier(0);
                if (next) {
                    if (i == 0) XS_DMA_W(ks + 1, 0)
                    if (i == 1) XS_DMA_W(ks + 1, 1)
                }
                if (slab_next) {                     // slot s of the next chunk's slab goes out in the K-step of context s % nctx
                    if (i == 2) { if (k_ci == 0) XS_SLAB(k_c + 1, 0) }
                    if (i == 3) {
                        const bool first = k_ci == 1;
                        if (first || k_ci == s2_ci) XS_SLAB_1OR2(k_c + 1, first)
                    }
                    if (i == (j == 1 ? 3 : 4)) { if (k_ci == s3_ci) XS_SLAB(k_c + 1, 3) }
                }
                if (j == 1) {                        // (issued LAST in this K-step: the next one waits for all but these)
                    if (i == 4) { XS_DMA_SW(ss, 0) XS_DMA_SW(ss, 1) }
                    if (i == 5) { XS_DMA_SW(ss, 2) XS_DMA_SW(ss, 3) }
                    if (i == 6) XS_DMA_SW(ss, 4)
                    if (i == 7) XS_DMA_SW(ss, 5)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // on to the next K-step's (chunk, context)
            if (ks + 1 < p.nk) {
                if (++k_ci == p.nctx) { k_ci = 0; ++k_c; }
            }
        }
        // M: the two block-scaled terms of this super-step. K block kb = lane quarter: its operand rows sit in the side slab of its
        // chunk at its context offset.
        __builtin_amdgcn_sched_barrier(0);
        {
            const bool m_live = 4 * ss + q4 < p.nk;
            const int mc = m_live ? m_c : p.nch_in - 1;       // (padded K blocks: zero weights, any resident slab)
            const int mo = XS_CTX(m_live ? m_ci : 0);
            int sa_b = XS_SS_OFF + (mc & 3) * XS_SSLAB + (arow0 + mo) * 16;          // e2m1 record of row block 0 (+ 256 B per block)
            int ss_b = XS_SS_OFF + (mc & 3) * XS_SSLAB + 2 * XS_SPLANE + (arow0 + mo) * 4;
            int sw_rec = sw_col;
            asm volatile("" : "+v"(sw_rec), "+v"(sa_b), "+v"(ss_b));
            const unsigned char* sW = rsm + XS_SW_OFF;
            u32x4 w4[4], wl6a[4];
            u32x2 wl6b[4];
            unsigned wsc[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int rec = sw_rec + jj * 16;
                w4[jj] = *reinterpret_cast<const u32x4*>(sW + rec * 16);
                wl6a[jj] = *reinterpret_cast<const u32x4*>(sW + 16384 + rec * 16);
                wl6b[jj] = *reinterpret_cast<const u32x2*>(sW + 32768 + rec * 8);
                wsc[jj] = *reinterpret_cast<const unsigned*>(sW + 40960 + rec * 4);
            }
            u32x4 l_n = *reinterpret_cast<const u32x4*>(rsm + sa_b);
            u32x4 h_n = *reinterpret_cast<const u32x4*>(rsm + sa_b + XS_SPLANE);
            unsigned s_n = *reinterpret_cast<const unsigned*>(rsm + ss_b);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x4 l = l_n, h = h_n;
                const unsigned asc = s_n;
                if (i < 7) {
                    l_n = *reinterpret_cast<const u32x4*>(rsm + sa_b + (i + 1) * 256);
                    h_n = *reinterpret_cast<const u32x4*>(rsm + sa_b + XS_SPLANE + (i + 1) * 256);
                    s_n = *reinterpret_cast<const unsigned*>(rsm + ss_b + (i + 1) * 64);
                }
                const i32x8 al = i32x8{(int)l.x, (int)l.y, (int)l.z, (int)l.w, 0, 0, 0, 0};
                const i32x8 ah = i32x8{(int)h.x, (int)h.y, (int)h.z, (int)h.w, 0, 0, 0, 0};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const i32x8 bw = i32x8{(int)w4[jj].x, (int)w4[jj].y, (int)w4[jj].z, (int)w4[jj].w, 0, 0, 0, 0};
                    acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al, bw, acc[i][jj], 4, 4, 0, asc, 0, wsc[jj]);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const i32x8 bw = i32x8{(int)wl6a[jj].x, (int)wl6a[jj].y, (int)wl6a[jj].z, (int)wl6a[jj].w, (int)wl6b[jj].x, (int)wl6b[jj].y, 0, 0};
                    acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah, bw, acc[i][jj], 4, 2, 1, asc, 1, wsc[jj]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // this lane's K block of the next super-step: four K-steps on
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4)
                if (++m_ci == p.nctx) { m_ci = 0; ++m_c; }
        }
    }
#undef XS_SLAB
#undef XS_SLAB_1OR2
#undef XS_SLAB_DMA
#undef XS_DMA_W
#undef XS_DMA_SW
#undef XS_CTX

#undef MX_PRM_OFF
#define MX_PRM_OFF XS_PRM_OFF
#include "tdnn_mx_epilogue.inc"
}

template <int ACT, int OUT>
__global__ __launch_bounds__(512) void tdnn_mxs_kernel(MxParams p, int mtiles, int ntiles, int gtiles, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    mxs_tile<ACT, OUT>(p, blockIdx.x, mtiles, ntiles, gtiles, stats, rsm);
}

// May this layer run on the slab kernel? (mx_launch, tdnn_mx.hip)
bool mxs_applies(const KtfTdnnDesc* d) {
    if (d->nctx < 2) return false;
    for (int i = 0; i < d->nctx; ++i)
        if (d->ctx[i] < -XS_PAD || d->ctx[i] > XS_PAD) return false;
    return true;
}

int mxs_launch(const MxParams& p, int64_t B, int act, int out_kind, double* stats, hipStream_t st) {
    const int mtiles = ktf_cdiv(p.T, 256), ntiles = ktf_cdiv(p.units, 256);
    const int64_t gtiles = B * mtiles;
    const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles;
#define XS_LAUNCH(A, O)                                                                                                \
    {                                                                                                                  \
        KTF_NOTE_KERNEL("tdnn_mxs_kernel");                                                                            \
        KTF_LDS_ONCE(XS_LDS_BYTES, tdnn_mxs_kernel<A, O>);                                                             \
        hipLaunchKernelGGL((tdnn_mxs_kernel<A, O>), dim3((unsigned)nblocks), dim3(512), XS_LDS_BYTES, st, p, mtiles, ntiles, (int)gtiles, stats); \
    }
    if (act == KTF_ACT_RELU) {
        if (out_kind == MX_OUT_STATS) XS_LAUNCH(KTF_ACT_RELU, MX_OUT_STATS) else if (out_kind == MX_OUT_F32) XS_LAUNCH(KTF_ACT_RELU, MX_OUT_F32) else XS_LAUNCH(KTF_ACT_RELU, MX_OUT_PLANES)
    } else {
        if (out_kind == MX_OUT_STATS) XS_LAUNCH(KTF_ACT_NONE, MX_OUT_STATS) else if (out_kind == MX_OUT_F32) XS_LAUNCH(KTF_ACT_NONE, MX_OUT_F32) else XS_LAUNCH(KTF_ACT_NONE, MX_OUT_PLANES)
    }
#undef XS_LAUNCH
    return KTF_OK;
}
