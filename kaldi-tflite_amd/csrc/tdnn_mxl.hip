// KTF_GEMM_F16MX on a 192 x 256 tile with DEDICATED LOADER WAVES: eight matrix waves (two per SIMD, 96 x 64 each: 6 x 4 MFMA tiles)
// issue nothing but LDS fragment reads and MFMAs; four loader waves (one per SIMD) issue every LDS-DMA of the tile. Same
// arithmetic as tdnn_mx.hip (one v_mfma_f32_16x16x32_f16 pass + the fp4 x fp4 and fp4 x fp6 terms on
// v_mfma_scale_f32_16x16x128_f8f6f4), same activation planes; the weights come as this kernel's own LDS images
// (mx.weight_images_loader).
//
// Why: in tdnn_mx.hip the eight waves that issue the MFMAs also issue the 224 LDS-DMA instructions of a super-step, and the two
// instruction streams add up (5.1 us = 3.0 us of MFMA issue + 224 x 9 ns) instead of overlapping: a wave that sits in vector-memory
// issue cannot issue the MFMAs behind it, and the 256 x 256 tile leaves no registers for more waves (docs/lab_notes_r3.md,
// tools/mx/overlap_probe.hip). Twelve waves fit at 168 registers with a 96 x 64 wave tile (96 accumulators).
//
// Schedule. A super-step (four 32-deep K-steps) is SIX phases of 24 MFMAs per matrix wave:
//     F0  Ma  F1  F2  Mb  F3
//   F0..F3  one half-precision K-step from a two-slot ring (slot = K-step & 1);
//   Ma, Mb  the block-scaled terms of the super-step for the wave's column blocks 0-1 / 2-3 (32 units each = one output chunk).
// One workgroup barrier opens every phase. The loader waves run one phase ahead of what they fill: during phase P they issue the
// DMAs of buffers whose last reader was phase P - 1 or earlier, then wait until everything issued BEFORE phase P has landed
// (`s_waitcnt vmcnt(n issued in P)`), so data issued in P is readable from phase P + 2:
//     F0: stage of F1 (slot 1)          Ma: stage of F2 (slot 0)          F1: W side half a of the next super-step
//     F2: stage of F3 (slot 1)          Mb: stage of the next F0 (slot 0) F3: W side half b of the next super-step
//   + the nine A-side pieces (per loader wave) of the NEXT super-step spread 1 1 2 1 1 3 over the phases (side A is double-buffered;
//     the W side is one buffer whose halves a / b are free again after Ma / Mb).
// Every phase carries 32 DMA instructions per CU (8 per loader wave) against 768 clk of MFMA issue per SIMD.
//
// LDS (160,768 B): ring 2 x (A 12 KiB | W 16 KiB) | side A 2 x 27 KiB | side W 44 KiB | epilogue constants 3 KiB.
// All images are FRAGMENT images: a wave reads an operand fragment as 64 lanes x 16 consecutive bytes (the A half stage keeps the
// 64-byte rows of tdnn_mx.hip with its XOR placement of the 16-byte pieces, so that the four lanes of a DMA that fetch one row
// stay adjacent).
//
// Rows. Plane / fp32 outputs: tiles walk the FLAT row space b * T + t, so a tile may span utterances (no padded rows between
// utterances: 998-frame utterances on 256-row tiles computed 2.6 % padding); the loader lanes clamp every row inside its own
// utterance (SAME padding = edge replication, layers/tdnn/tdnn.py:246-247). Fused pooling: tiles stay inside one utterance, so the
// partial sums of an utterance do not depend on what else is in the batch (batch == single bit for bit); row blocks without a
// valid row issue no MFMAs, and since the two matrix waves of a SIMD are the two row halves of the tile, a tile whose lower half
// is empty runs at the other half's full matrix rate.
//
// Epilogue (planes): the MFMA operands are SWAPPED for the plane / fp32 outputs (weights as the A operand), so a lane holds, for
// frame r16 of a row block, eight consecutive units (the unit order inside a 32-unit chunk is permuted in the weight images): the
// 32-value MX blocks are encoded from registers -- two cross-lane maxima over the four lanes of a frame (v_permlane16_swap /
// v_permlane32_swap), no LDS staging, no barriers -- and a half piece leaves as 16 bytes per lane, 1 KiB of consecutive records
// per store instruction.
//
// Replaces: layers/tdnn/tdnn.py:251-280 (+ keras ReLU, batchnorm.py:78-88, stats_pooling.py:211-240 when fused).
#include "tdnn_mx_common.h"

#define XL_ROWS 192
#define XL_STAGE_A 12288                             // A image of a K-step: 12 row blocks x (16 rows x 64 B)
#define XL_STAGE (XL_STAGE_A + 16384)                // ... | W image: 16 unit-block fragments x 1 KiB
#define XL_SA_OFF (2 * XL_STAGE)                     // side A (x 2): xl4 12 x 1 KiB | x4 12 x 1 KiB | scales 12 x 256 B
#define XL_SA_BYTES (2 * 12288 + 3072)
#define XL_SW_OFF (XL_SA_OFF + 2 * XL_SA_BYTES)      // side W: two halves of w4 8 x 1 KiB | wl6a 8 x 1 KiB | wl6b 8 x 512 B | scales 8 x 256 B
#define XL_SWH 22528
#define XL_WQ_BLOCK (2 * XL_SWH)                     // = bytes of one (N-tile, super-step) block of mx.weight_images_loader's wq
#define XL_PRM_OFF (XL_SW_OFF + XL_WQ_BLOCK)
#define XL_LDS_BYTES (XL_PRM_OFF + 3 * 256 * 4)      // 160,768 B

static_assert(XL_LDS_BYTES <= 163840, "LDS budget");

struct MxlParams {
    MxParams m;
    double* stats;
    int32_t B, mtiles, ntiles, gtiles;
    uint32_t total_rows;                             // B * T
};

__device__ __forceinline__ float xl_act(float v, int act) { return act == KTF_ACT_RELU ? fmaxf(v, 0.0f) : v; }

#define XL_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
#define XL_BARRIER()                                                                                                   \
    {                                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
    }

template <int ACT, int OUT>
__global__ __launch_bounds__(768) void tdnn_mxl_kernel(MxlParams q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    constexpr bool FLAT = OUT != MX_OUT_STATS;       // tiles over the flat row space (may span utterances)
    constexpr bool SWAP = OUT != MX_OUT_STATS;       // weights as the A operand of the MFMAs
    const MxParams& p = q.m;
    const int id = blockIdx.x;
    const int xcd = id & 7, gslot = id >> 3;         // an XCD runs all N-tiles of an M-tile back to back (its L2 keeps the A rows)
    const int g = (gslot / q.ntiles) * 8 + xcd;
    const int nt = gslot - (gslot / q.ntiles) * q.ntiles;
    if (g >= q.gtiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (int)p.T;
    // ---- the tile's rows
    int b0, t0;
    unsigned f0;
    if constexpr (FLAT) {
        f0 = (unsigned)g * XL_ROWS;
        b0 = (int)(f0 / (unsigned)T);
        t0 = (int)(f0 - (unsigned)b0 * (unsigned)T);
    } else {
        b0 = g / q.mtiles;
        t0 = (g - b0 * q.mtiles) * XL_ROWS;
        f0 = (unsigned)b0 * (unsigned)T + (unsigned)t0;
    }
    int vend = 0;                                    // rows [vend, 192) of the tile hold no valid row
    if constexpr (FLAT) {
        int b = b0;
        long long fb = (long long)b0 * T;
        while (fb < (long long)f0 + XL_ROWS && b < q.B) {
            const int len = p.lens ? p.lens[b] : T;
            const long long lo = fb > (long long)f0 ? fb : (long long)f0;
            long long hi = fb + len;
            if (hi > (long long)f0 + XL_ROWS) hi = (long long)f0 + XL_ROWS;
            if (hi > lo) vend = (int)(hi - (long long)f0);
            ++b;
            fb += T;
        }
    } else {
        const int len = p.lens ? p.lens[b0] : T;
        vend = len - t0 < XL_ROWS ? len - t0 : XL_ROWS;
    }
    if (vend <= 0) return;

    const int nkp = p.nss * 4;
    const char* whn = p.wh + (int64_t)nt * nkp * 16384;
    const char* wqn = p.wq + (int64_t)nt * p.nss * XL_WQ_BLOCK;
    const unsigned long long cpk0 = p.ctx_pk[0], cpk1 = p.ctx_pk[1];

    if (wave >= 8) {
        // ======================================================================================= loader wave l
        const int l = wave - 8;
        // rows this lane fetches: half stages (16 rows x 64 B per piece: row = lane >> 2, 16-byte position lane & 3) and side pieces
        // (row = lane & 15 of the row block, K block lane >> 4), row blocks l, l + 4, l + 8
        unsigned hb[3], sb[3];                       // first record of the row's utterance: b * nch_in * T
        int ht[3], hl[3], st[3], sl_[3];             // the row's frame index, and len - 1 of its utterance (clamp bound)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int row = (l + 4 * k) * 16 + (w ? (lane & 15) : (lane >> 2));
                int b, t;
                if constexpr (FLAT) {
                    unsigned f = f0 + (unsigned)row;
                    if (f >= q.total_rows) f = q.total_rows - 1;
                    b = (int)(f / (unsigned)T);
                    t = (int)(f - (unsigned)b * (unsigned)T);
                } else {
                    b = b0;
                    t = t0 + row;
                }
                const int len = p.lens ? p.lens[b] : T;
                const unsigned base = (unsigned)(b - b0) * (unsigned)(p.nch_in * T);       // relative to utterance b0 (32-bit byte offsets below)
                const int lm1 = len > 0 ? len - 1 : 0;
                if (w) { sb[k] = base; st[k] = t; sl_[k] = lm1; } else { hb[k] = base; ht[k] = t; hl[k] = lm1; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the lens loads: from here on the vector-memory counter counts DMAs only)
        // the planes from the first record of utterance b0 on: uniform bases + 32-bit per-lane byte offsets (a tile spans at most
        // 192 + T rows: offsets stay below 2^32 while T * din_pad * 2 < 2^31, which the launcher requires)
        const uint64_t u0 = (uint64_t)b0 * (uint64_t)(p.nch_in * T);
        const char* xh0 = p.xh + u0 * 64u;
        const char* xl0 = p.xl4 + u0 * 16u;
        const char* x40 = p.x4 + u0 * 16u;
        const char* xs0 = p.xs + u0 * 4u;
        const unsigned hpos = (unsigned)((((lane & 3) ^ ((4 - (((lane >> 2) >> 2) & 3)) & 3)) * 16));   // this lane's 16-byte piece of its row
        // half stages walk the K-steps in order: context index, offset, first record of the chunk (chunk * T)
        int h_ci = 0;
        unsigned h_cT = 0;
#define XL_CTX(ci_) ((int)(signed char)(((ci_) < 8 ? cpk0 : cpk1) >> (((ci_) & 7) * 8)))
        int h_off = XL_CTX(0);
        // one half stage: 3 row pieces + 4 weight pieces per loader wave, then on to the next K-step (padded K-steps re-read step 0)
#define XL_H(ks_, slot_)                                                                                               \
        {                                                                                                              \
            unsigned char* st__ = rsm + (slot_) * XL_STAGE;                                                            \
            _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                                         \
                const unsigned vo_ = (unsigned)(ks_) * 16384u + (unsigned)((l + 4 * k_) * 1024 + lane * 16);           \
                __builtin_amdgcn_global_load_lds((glb_ptr_t*)(whn + vo_), (lds_ptr_t*)(st__ + XL_STAGE_A + (l + 4 * k_) * 1024), 16, 0, 0); \
            }                                                                                                          \
            _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                                         \
                int r_ = ht[k_] + h_off;                                                                               \
                r_ = r_ < 0 ? 0 : (r_ > hl[k_] ? hl[k_] : r_);                                                         \
                const unsigned vo_ = (hb[k_] + h_cT + (unsigned)r_) * 64u + hpos;                                      \
                __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xh0 + vo_), (lds_ptr_t*)(st__ + (l + 4 * k_) * 1024), 16, 0, 0); \
            }                                                                                                          \
            if ((ks_) + 1 < p.nk) {                                                                                    \
                if (++h_ci == p.nctx) { h_ci = 0; h_cT += (unsigned)T; }                                               \
                h_off = XL_CTX(h_ci);                                                                                  \
            } else {                                                                                                   \
                h_cT = 0;                                                                                              \
                h_off = XL_CTX(0);                                                                                     \
            }                                                                                                          \
        }
        // side A: this lane's K block is lane >> 4; its K-step of the super-step being fetched: chunk base and offset (per lane)
        const int kb = lane >> 4;
        int s_ci = kb % p.nctx;
        unsigned s_cT = (unsigned)(kb / p.nctx) * (unsigned)T;
        int s_ks = kb;
        int s_off;
#define XL_S_SET()                                                                                                     \
        {                                                                                                              \
            const bool live_ = s_ks < p.nk;                                                                            \
            const int ci_ = live_ ? s_ci : 0;                                                                          \
            s_off = (int)(signed char)((ci_ < 8 ? cpk0 : cpk1) >> ((ci_ & 7) * 8));                                    \
        }
#define XL_S_ADV()                                                                                                     \
        {                                                                                                              \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                         \
                if (++s_ci == p.nctx) { s_ci = 0; s_cT += (unsigned)T; }                                               \
            }                                                                                                          \
            s_ks += 4;                                                                                                 \
            XL_S_SET()                                                                                                 \
        }
        XL_S_SET()
        // piece j_ = 0..8 of this wave's share of side A (row block l + 4 (j_ % 3); j_ / 3: residual codes, value codes, scale words)
#define XL_SA(buf_, j_)                                                                                                \
        {                                                                                                              \
            constexpr int k_ = (j_) % 3, kind_ = (j_) / 3;                                                             \
            unsigned char* sa__ = rsm + XL_SA_OFF + (buf_) * XL_SA_BYTES;                                              \
            int r_ = st[k_] + s_off;                                                                                   \
            r_ = r_ < 0 ? 0 : (r_ > sl_[k_] ? sl_[k_] : r_);                                                           \
            const unsigned rec_ = sb[k_] + (s_ks < p.nk ? s_cT : 0u) + (unsigned)r_;                                   \
            if (kind_ == 0)                                                                                            \
                __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xl0 + rec_ * 16u), (lds_ptr_t*)(sa__ + (l + 4 * k_) * 1024), 16, 0, 0); \
            else if (kind_ == 1)                                                                                       \
                __builtin_amdgcn_global_load_lds((glb_ptr_t*)(x40 + rec_ * 16u), (lds_ptr_t*)(sa__ + 12288 + (l + 4 * k_) * 1024), 16, 0, 0); \
            else                                                                                                       \
                __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xs0 + rec_ * 4u), (lds_ptr_t*)(sa__ + 24576 + (l + 4 * k_) * 256), 4, 0, 0); \
        }
        // side W of super-step ss_, half h_: pieces l, l + 4, ... of its 22 KiB
#define XL_SW(ss_, h_)                                                                                                 \
        {                                                                                                              \
            _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) {                                                         \
                const int pc_ = l + 4 * k_;                                                                            \
                if (pc_ < 22) {                                                                                        \
                    const unsigned vo_ = (unsigned)(ss_) * (unsigned)XL_WQ_BLOCK + (unsigned)((h_) * XL_SWH + pc_ * 1024 + lane * 16); \
                    __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wqn + vo_), (lds_ptr_t*)(rsm + XL_SW_OFF + (h_) * XL_SWH + pc_ * 1024), 16, 0, 0); \
                }                                                                                                      \
            }                                                                                                          \
        }
        // ---- prologue: the first half stage, then both W side halves and side A of super-step 0
        XL_H(0, 0)
        XL_SW(0, 0)
        XL_SW(0, 1)
        XL_SA(0, 0) XL_SA(0, 1) XL_SA(0, 2) XL_SA(0, 3) XL_SA(0, 4) XL_SA(0, 5) XL_SA(0, 6) XL_SA(0, 7) XL_SA(0, 8)
        XL_S_ADV()
        XL_WAIT_VM(19);                              // the 7 DMAs of the stage have landed (19 or 21 were issued behind them)
        XL_BARRIER()                                 // opens F0 of super-step 0
        for (int ss = 0; ss < p.nss; ++ss) {
            const int ks = 4 * ss;
            const int nb = (ss + 1) & 1;             // side A buffer of the next super-step
            if (ss + 1 < p.nss) {
                XL_H(ks + 1, 1) XL_SA(nb, 0)                          XL_WAIT_VM(8); XL_BARRIER()      // during F0; opens Ma
                XL_H(ks + 2, 0) XL_SA(nb, 1)                          XL_WAIT_VM(8); XL_BARRIER()      // during Ma; opens F1
                XL_SW(ss + 1, 0) XL_SA(nb, 2) XL_SA(nb, 3)            XL_WAIT_VM(7); XL_BARRIER()      // during F1; opens F2
                XL_H(ks + 3, 1) XL_SA(nb, 4)                          XL_WAIT_VM(8); XL_BARRIER()      // during F2; opens Mb
                XL_H(ks + 4, 0) XL_SA(nb, 5)                          XL_WAIT_VM(8); XL_BARRIER()      // during Mb; opens F3
                XL_SW(ss + 1, 1) XL_SA(nb, 6) XL_SA(nb, 7) XL_SA(nb, 8)
                XL_S_ADV()
                XL_WAIT_VM(8); XL_BARRIER()                                                           // during F3; opens the next F0
            } else {
                XL_H(ks + 1, 1) XL_WAIT_VM(7); XL_BARRIER()
                XL_H(ks + 2, 0) XL_WAIT_VM(7); XL_BARRIER()
                XL_WAIT_VM(0); XL_BARRIER()
                XL_H(ks + 3, 1) XL_WAIT_VM(7); XL_BARRIER()
                XL_WAIT_VM(0); XL_BARRIER()
            }
        }
#undef XL_H
#undef XL_SA
#undef XL_SW
#undef XL_S_SET
#undef XL_S_ADV
#undef XL_CTX
        return;
    }

    // =========================================================================================== matrix wave (wm, wn)
    const int wm = wave >> 2, wn = wave & 3;
    int nblk = (vend - wm * 96 + 15) >> 4;           // row blocks of this wave that hold a valid row
    nblk = nblk < 0 ? 0 : (nblk > 6 ? 6 : nblk);
    const int n0 = nt * 256;
    if (tid < 256) {                                 // epilogue constants of the tile's units (visible behind the first barrier)
        float* prm = reinterpret_cast<float*>(rsm + XL_PRM_OFF);
        const int n = n0 + tid;
        const bool nv = n < p.units;
        prm[tid] = (nv && p.bias) ? p.bias[n] : 0.0f;
        prm[256 + tid] = (nv && p.scale) ? p.scale[n] : 1.0f;
        prm[512 + tid] = (nv && p.shift) ? p.shift[n] : 0.0f;
    }
    f32x4 acc[6][4];
    const int r16 = lane & 15, q4 = lane >> 4;
    const int fr = (4 - ((r16 >> 2) & 3)) & 3;
    const int a_off = wm * 6144 + r16 * 64 + ((q4 ^ fr) << 4);        // A half fragment of row block 0 (+ 1 KiB per block)
    const int b_off = XL_STAGE_A + wn * 4096 + lane * 16;             // W half fragment of unit block 0 (+ 1 KiB per block)

    // one half-precision K-step from ring slot `slot_`
#define XL_F(slot_)                                                                                                    \
    {                                                                                                                  \
        const unsigned char* st__ = rsm + (slot_) * XL_STAGE;                                                          \
        hfrag8 bh[4];                                                                                                  \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) bh[jj] = *reinterpret_cast<const hfrag8*>(st__ + b_off + jj * 1024); \
        hfrag8 a_cur = *reinterpret_cast<const hfrag8*>(st__ + a_off);                                                 \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                \
            if (i < nblk) {                                                                                            \
                hfrag8 a_nxt = a_cur;                                                                                  \
                if (i < 5) a_nxt = *reinterpret_cast<const hfrag8*>(st__ + a_off + (i + 1) * 1024);                    \
                _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                       \
                    acc[i][jj] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[jj], a_cur, acc[i][jj], 0, 0, 0)     \
                                      : __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);    \
                a_cur = a_nxt;                                                                                         \
            }                                                                                                          \
        }                                                                                                              \
    }
    // the two block-scaled terms of the super-step for unit blocks 2 h_, 2 h_ + 1 (side A buffer buf_)
#define XL_M(buf_, h_)                                                                                                 \
    {                                                                                                                  \
        const unsigned char* sA = rsm + XL_SA_OFF + (buf_) * XL_SA_BYTES + wm * 6144;                                  \
        const unsigned char* sW = rsm + XL_SW_OFF + (h_) * XL_SWH;                                                     \
        u32x4 w4[2], wl6a[2];                                                                                          \
        u32x2 wl6b[2];                                                                                                 \
        unsigned wsc[2];                                                                                               \
        _Pragma("unroll") for (int jl = 0; jl < 2; ++jl) {                                                             \
            const int cbh = wn * 2 + jl;                                                                               \
            w4[jl] = *reinterpret_cast<const u32x4*>(sW + cbh * 1024 + lane * 16);                                     \
            wl6a[jl] = *reinterpret_cast<const u32x4*>(sW + 8192 + cbh * 1024 + lane * 16);                            \
            wl6b[jl] = *reinterpret_cast<const u32x2*>(sW + 16384 + cbh * 512 + lane * 8);                             \
            wsc[jl] = *reinterpret_cast<const unsigned*>(sW + 20480 + cbh * 256 + lane * 4);                           \
        }                                                                                                              \
        u32x4 l_n = *reinterpret_cast<const u32x4*>(sA + lane * 16);                                                   \
        u32x4 h_n = *reinterpret_cast<const u32x4*>(sA + 12288 + lane * 16);                                           \
        unsigned s_n = *reinterpret_cast<const unsigned*>(sA + 24576 - wm * 6144 + wm * 1536 + lane * 4);              \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                \
            if (i < nblk) {                                                                                            \
                const u32x4 lq = l_n, hq = h_n;                                                                        \
                const unsigned asc = s_n;                                                                              \
                if (i < 5) {                                                                                           \
                    l_n = *reinterpret_cast<const u32x4*>(sA + (i + 1) * 1024 + lane * 16);                            \
                    h_n = *reinterpret_cast<const u32x4*>(sA + 12288 + (i + 1) * 1024 + lane * 16);                    \
                    s_n = *reinterpret_cast<const unsigned*>(sA + 24576 - wm * 6144 + wm * 1536 + (i + 1) * 256 + lane * 4); \
                }                                                                                                      \
                const i32x8 al = i32x8{(int)lq.x, (int)lq.y, (int)lq.z, (int)lq.w, 0, 0, 0, 0};                        \
                const i32x8 ah = i32x8{(int)hq.x, (int)hq.y, (int)hq.z, (int)hq.w, 0, 0, 0, 0};                        \
                _Pragma("unroll") for (int jl = 0; jl < 2; ++jl) {   /* fp4 residual of x times the fp4 image of w (scale bytes 0) */ \
                    const i32x8 bw = i32x8{(int)w4[jl].x, (int)w4[jl].y, (int)w4[jl].z, (int)w4[jl].w, 0, 0, 0, 0};    \
                    acc[i][2 * (h_) + jl] = SWAP                                                                       \
                        ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, al, acc[i][2 * (h_) + jl], 4, 4, 0, wsc[jl], 0, asc) \
                        : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al, bw, acc[i][2 * (h_) + jl], 4, 4, 0, asc, 0, wsc[jl]); \
                }                                                                                                      \
                _Pragma("unroll") for (int jl = 0; jl < 2; ++jl) {   /* fp4 image of x times the fp6 residual of w (scale bytes 1) */ \
                    const i32x8 bw = i32x8{(int)wl6a[jl].x, (int)wl6a[jl].y, (int)wl6a[jl].z, (int)wl6a[jl].w, (int)wl6b[jl].x, (int)wl6b[jl].y, 0, 0}; \
                    acc[i][2 * (h_) + jl] = SWAP                                                                       \
                        ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, ah, acc[i][2 * (h_) + jl], 2, 4, 1, wsc[jl], 1, asc) \
                        : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah, bw, acc[i][2 * (h_) + jl], 4, 2, 1, asc, 1, wsc[jl]); \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
    // a phase boundary: this wave's fragment reads are complete (their buffers may be refilled behind the barrier)
#define XL_PHASE()                                                                                                     \
    {                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        XL_BARRIER()                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
    const float* prm = reinterpret_cast<const float*>(rsm + XL_PRM_OFF);
    XL_PHASE()                                       // opens F0 of super-step 0; the epilogue constants are in place
    // the accumulators start at the bias of their units
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b4;
        if constexpr (SWAP) {
            b4 = *reinterpret_cast<const f32x4*>(prm + mx_unit(wn * 4 + j, q4 * 4));
        } else {
            const float bv = prm[mx_unit(wn * 4 + j, r16)];
            b4 = f32x4{bv, bv, bv, bv};
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i][j] = b4;
    }
    for (int ss = 0; ss < p.nss; ++ss) {
        const int buf = ss & 1;
        XL_F(0)
        XL_PHASE() XL_M(buf, 0)
        XL_PHASE() XL_F(1)
        XL_PHASE() XL_F(0)
        XL_PHASE() XL_M(buf, 1)
        XL_PHASE() XL_F(1)
        if (ss + 1 < p.nss) XL_PHASE()
    }
    __builtin_amdgcn_sched_barrier(0);
#undef XL_F
#undef XL_M
#undef XL_PHASE

    if constexpr (OUT == MX_OUT_STATS) {
        // fused StatsPooling (stats_pooling.py:231-240): per unit the sum and the sum of squares of the wave's rows, in fp32 relative
        // to a pivot (row 0 of the wave's block: a constant column -- a dead ReLU unit -- gives exactly 0 and 0), then fp64.
        // Accumulator (i, j)[r] = row 16 i + 4 q4 + r, unit block j, column r16.
        const int rv = vend - wm * 96;               // valid rows of this wave's block
        if (rv <= 0) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ul = mx_unit(wn * 4 + j, r16);
            const float esc = prm[256 + ul], esh = prm[512 + ul];
            const float v0 = xl_act(acc[0][j][0], ACT) * esc + esh;
            const float pv = __shfl(v0, lane & 15, 64);
            float s32 = 0.0f, q32 = 0.0f;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = xl_act(acc[i][j][r], ACT) * esc + esh;
                    if (rv >= 96 || i * 16 + q4 * 4 + r < rv) {
                        const float u = v - pv;
                        s32 += u;
                        q32 = fmaf(u, u, q32);
                        ++cnt;
                    }
                }
            }
            const double pd = (double)pv, sd = (double)s32, nd = (double)cnt;
            double s = sd + nd * pd;
            double qq = (double)q32 + 2.0 * pd * sd + nd * pd * pd;
            s += __shfl_xor(s, 16, 64); qq += __shfl_xor(qq, 16, 64);
            s += __shfl_xor(s, 32, 64); qq += __shfl_xor(qq, 32, 64);
            const int n = n0 + ul;
            if (lane < 16 && n < p.units) {
                if (p.stat_slots > 0) {              // one slot per 96-row block of the utterance, written by exactly one wave
                    double* dst = q.stats + (((int64_t)b0 * p.stat_slots + (t0 / 96 + wm)) * 2) * p.units + n;
                    dst[0] = s;
                    dst[p.units] = qq;
                } else {
                    double* dst = q.stats + ((int64_t)b0 * 2) * p.units + n;
                    atomicAdd(dst, s);
                    atomicAdd(dst + p.units, qq);
                }
            }
        }
        return;
    } else {
        // Accumulator (i, jj)[r] = frame 16 i + r16 of the wave's rows, unit mx_unit(4 wn + jj, 4 q4 + r): for output chunk c of the
        // wave (unit blocks 2 c, 2 c + 1) this lane holds units 8 q4 .. 8 q4 + 7 of the chunk.
        // the rows: (utterance, frame) of this lane's frame in every row block, and whether it is a valid row
        unsigned rec_i[6];                           // planes: record of (b, chunk 0, t): b * nch_out * T + t; fp32 rows: b * T + t
        unsigned ok = 0;                             // bit i: row block i's frame is valid
        {
            int b, t;
            const int row0 = wm * 96 + r16;
            if constexpr (FLAT) {
                const unsigned f = f0 + (unsigned)row0;
                b = (int)(f / (unsigned)T);
                t = (int)(f - (unsigned)b * (unsigned)T);
            } else {
                b = b0;
                t = t0 + row0;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                while (FLAT && t >= T) { t -= T; ++b; }
                const bool inb = b < q.B;
                const int len = inb ? (p.lens ? p.lens[b] : T) : 0;
                if (i < nblk && t < len) ok |= 1u << i;
                rec_i[i] = (unsigned)(b - b0) * (unsigned)(OUT == MX_OUT_PLANES ? p.nch_out * T : T) + (unsigned)t;   // relative to utterance b0
                t += 16;
            }
        }
        const bool affine = p.scale != nullptr;
        if constexpr (OUT == MX_OUT_PLANES) {
            // per chunk: the half pieces leave per row block (16 bytes per lane: 1 KiB of consecutive records per store); the e2m1
            // dwords and the scale words of row blocks 0-3 are transposed between register index and lane quarter so that lane quarter
            // q stores the whole 16-byte record of row block q (row = lane of the wave's block), those of row blocks 4, 5 pairwise
            // (8 bytes per lane): 12 store instructions per chunk instead of 24
            const uint64_t o0 = (uint64_t)b0 * (uint64_t)(p.nch_out * T);        // uniform bases + 32-bit per-lane byte offsets
            char* yh0 = p.yh + o0 * 64u;
            char* yl0 = p.yl4 + o0 * 16u;
            char* y40 = p.y4 + o0 * 16u;
            char* ys0 = p.ys + o0 * 4u;
            const unsigned recA = q4 == 0 ? rec_i[0] : q4 == 1 ? rec_i[1] : q4 == 2 ? rec_i[2] : rec_i[3];
            const unsigned recB = (q4 & 1) ? rec_i[5] : rec_i[4];
            const bool okA = (ok >> q4) & 1u, okB = (ok >> (4 + (q4 & 1))) & 1u;
            // (no BatchNorm affine on this path: a plane output feeds a layer of the route, which folds it into its weights; the
            // launcher refuses scale / shift here -- a second, affine copy of this epilogue cost registers: spills whose reloads wait
            // on `vmcnt`, i.e. on the plane stores in flight)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int chunk = (n0 >> 5) + wn * 2 + c;
                if (chunk >= p.nch_out) continue;    // (wave-uniform)
                const unsigned crec = (unsigned)chunk * (unsigned)T;
                unsigned l4r[6], h4r[6], swr[6];
#pragma unroll
                for (int i2 = 0; i2 < 6; i2 += 2) {
                    float v[2][8];
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int e = 0; e < 8; ++e)      // the planes saturate at the largest half: one v_med3 does the ReLU and the clamp
                            v[n][e] = __builtin_amdgcn_fmed3f(acc[i2 + n][2 * c + (e >> 2)][e & 3], ACT == KTF_ACT_RELU ? 0.0f : -65504.0f, 65504.0f);
                    u32x4 hp[2];
                    unsigned l4p[2], h4p[2], swp[2];
                    mx_encode8<2>(v, hp, l4p, h4p, swp);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int i = i2 + n;
                        l4r[i] = l4p[n]; h4r[i] = h4p[n]; swr[i] = swp[n];
                        if ((ok >> i) & 1u) {
                            unsigned rr = rec_i[i];
                            asm volatile("" : "+v"(rr));      // (the address is formed here, not hoisted and kept in registers)
                            __builtin_nontemporal_store(hp[n], reinterpret_cast<u32x4*>(yh0 + ((rr + crec) * 64u + (unsigned)q4 * 16u)));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                mx_transpose4(l4r[0], l4r[1], l4r[2], l4r[3]);
                mx_transpose4(h4r[0], h4r[1], h4r[2], h4r[3]);
                if (okA) {
                    const unsigned rec = recA + crec;
                    const unsigned sw = q4 == 0 ? swr[0] : q4 == 1 ? swr[1] : q4 == 2 ? swr[2] : swr[3];
                    __builtin_nontemporal_store(u32x4{l4r[0], l4r[1], l4r[2], l4r[3]}, reinterpret_cast<u32x4*>(yl0 + rec * 16u));
                    __builtin_nontemporal_store(u32x4{h4r[0], h4r[1], h4r[2], h4r[3]}, reinterpret_cast<u32x4*>(y40 + rec * 16u));
                    __builtin_nontemporal_store(sw, reinterpret_cast<unsigned*>(ys0 + rec * 4u));
                }
                {   // row blocks 4, 5: even lane quarters end up with dwords (q, q + 1) of block 4, odd ones with (q - 1, q) of block 5
                    auto r = __builtin_amdgcn_permlane16_swap(l4r[4], l4r[5], false, false);
                    const u32x2 lw = u32x2{r[0], r[1]};
                    r = __builtin_amdgcn_permlane16_swap(h4r[4], h4r[5], false, false);
                    const u32x2 hw2 = u32x2{r[0], r[1]};
                    if (okB) {
                        const unsigned rec = recB + crec;
                        __builtin_nontemporal_store(lw, reinterpret_cast<u32x2*>(yl0 + (rec * 16u + (unsigned)(q4 >> 1) * 8u)));
                        __builtin_nontemporal_store(hw2, reinterpret_cast<u32x2*>(y40 + (rec * 16u + (unsigned)(q4 >> 1) * 8u)));
                        if (q4 < 2) __builtin_nontemporal_store((q4 & 1) ? swr[5] : swr[4], reinterpret_cast<unsigned*>(ys0 + rec * 4u));
                    }
                }
            }
        } else {
            // fp32 rows (B, T, ldy): four consecutive units per accumulator
            const bool vec = (p.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(p.yf) & 15) == 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int ul = mx_unit(wn * 4 + jj, q4 * 4);
                const int n = n0 + ul;
                float es[4], eh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { es[e] = prm[256 + ul + e]; eh[e] = prm[512 + ul + e]; }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    if (!((ok >> i) & 1u)) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = xl_act(acc[i][jj][e], ACT);
                        if (affine) v[e] = v[e] * es[e] + eh[e];
                    }
                    float* yp = p.yf + ((int64_t)b0 * T + (int64_t)rec_i[i]) * p.ldy + n;
                    if (vec && n + 4 <= p.units) {
                        *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.units) yp[e] = v[e];
                    }
                }
            }
        }
    }
}

int mxl_launch(const MxParams& p, int64_t B, int act, int out_kind, double* stats, hipStream_t st) {
    MxlParams q;
    memset(&q, 0, sizeof(q));
    q.m = p;
    q.stats = stats;
    q.B = (int32_t)B;
    KTF_REQUIRE(!(out_kind == MX_OUT_PLANES && p.scale), "ktf_tdnn_mx: the loader kernel writes planes without scale / shift (fold the BatchNorm into the "
                "next layer, or clear KTF_TDNN_MX_LOADER)");
    KTF_REQUIRE(B * p.T < (1ll << 31) && B * p.T * (int64_t)(p.nch_in > p.nch_out ? p.nch_in : p.nch_out) < (1ll << 32),
                "ktf_tdnn_mx: batch too large for the loader kernel's 32-bit record indices (B * T * D / 32 must be below 2^32)");
    q.total_rows = (uint32_t)(B * p.T);
    q.ntiles = ktf_cdiv(p.units, 256);
    if (out_kind == MX_OUT_STATS) {
        q.mtiles = ktf_cdiv(p.T, XL_ROWS);
        q.gtiles = (int32_t)(B * q.mtiles);
    } else {
        q.mtiles = 0;
        q.gtiles = ktf_cdiv(B * p.T, XL_ROWS);
    }
    const int64_t nblocks = (((int64_t)q.gtiles + 7) / 8) * 8 * q.ntiles;
#define XL_LAUNCH(A, O)                                                                                                \
    {                                                                                                                  \
        KTF_NOTE_KERNEL("tdnn_mxl_kernel");                                                                            \
        KTF_LDS_ONCE(XL_LDS_BYTES, tdnn_mxl_kernel<A, O>);                                                             \
        hipLaunchKernelGGL((tdnn_mxl_kernel<A, O>), dim3((unsigned)nblocks), dim3(768), XL_LDS_BYTES, st, q);          \
    }
    if (act == KTF_ACT_RELU) {
        if (out_kind == MX_OUT_STATS) XL_LAUNCH(KTF_ACT_RELU, MX_OUT_STATS) else if (out_kind == MX_OUT_F32) XL_LAUNCH(KTF_ACT_RELU, MX_OUT_F32) else XL_LAUNCH(KTF_ACT_RELU, MX_OUT_PLANES)
    } else {
        if (out_kind == MX_OUT_STATS) XL_LAUNCH(KTF_ACT_NONE, MX_OUT_STATS) else if (out_kind == MX_OUT_F32) XL_LAUNCH(KTF_ACT_NONE, MX_OUT_F32) else XL_LAUNCH(KTF_ACT_NONE, MX_OUT_PLANES)
    }
#undef XL_LAUNCH
    return KTF_OK;
}
