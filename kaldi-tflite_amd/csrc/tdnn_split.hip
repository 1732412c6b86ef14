// The plane kernels of KTF_GEMM_BF16X3 (split-bf16: x = hi + lo, w = hi + lo, three bf16 MFMA passes, fp32-grade accuracy) on the
// 256 x 256 ring tile.
#include "tdnn_ring.h"
#include "flat_stats.h"

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

// (Activations that already travel as hi / lo bf16 planes -- the model's own route -- run on tdnn_x3s_kernel below, whose K-loop
// carries no conversion: the in-register split here costs ~190 VALU instructions per wave per K-step, four times redundantly
// per A tile. This kernel serves callers that hand in fp32 activations.)
template <int ACT, bool STATS>
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

    constexpr int XB = 4;                              // bytes per activation element
    const char* xb = reinterpret_cast<const char*>(p.x) + ((int64_t)b * p.T * p.ldx) * XB;
    const char* wb = reinterpret_cast<const char*>(p.w);
    const char* wl = reinterpret_cast<const char*>(p.w_lo);
    const unsigned ldxb = (unsigned)p.ldx * XB;

    // A staging: chunk q = i*512 + tid (i < 4) -> row q/8, LDS position q%8, global chunk (q%8) ^ ((row>>1)&7)
    int a_t[4];
    unsigned a_cb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 512 + tid;
        const int row = q >> 3;
        a_cb[i] = (unsigned)(((q & 7) ^ ((row >> 1) & 7)) * 16);
        a_t[i] = start + (t0 + row) * p.sub;
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
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xb + vo_), (lds_ptr_t*)(st_ + i * 8192), 16, 0, 0); \
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
    const int a_row_off = (wm * 128 + (lane & 31)) * 128;
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
            {
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
// The plane kernel: hi / lo activation planes, hi / lo weight planes, acc += hi*hi + lo*hi + hi*lo on
// v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on this shape than on 32x32x16): 256 x 256 tile, 8 waves of 128 x 64, a
// stage = A hi | A lo | W hi | W lo (4 x 16 KiB, the chunk permutation of the 16x16x32 bf16 kernel), double buffered.
// Hand-scheduled K-step: the stage's operand DMAs are not issued in one burst behind the barrier (all eight waves then sit in
// DMA issue and LDS latency together while the matrix pipes idle) but one at a time between groups of MFMAs, and the A fragments
// of row group g+1 are read while the MFMAs of group g run (two fragment register sets).
#define XS_STAGE_BYTES (4 * R_TILE_BYTES)                // 64 KiB
#define XS_LDS_BYTES (2 * XS_STAGE_BYTES)                // 128 KiB
// SKIP: row groups (two 16-row blocks) of the wave's 128 rows that hold no valid output row issue no MFMAs -- for launches whose
// tiles are mostly padding (1.5 s windows: 148 rows of a 256-row tile, -17 % per step; the launcher decides from T). On full tiles the
// test costs 1.7 % (it splits the scheduler's K-step into regions), so the plain instantiation keeps them.
// FLAT (without SKIP: flat tiles are full but for the batch's last one, whose rows beyond the end are computed and not stored; SAME padding,
// no subsampling): the M-tiles cover the batch's VALID rows laid end
// to end (p.row_starts: exclusive prefix sums of lens) instead of 256-row tiles per utterance -- a 1.5 s window is 148 rows, 0.58 of a
// tile. A tile's rows belong to several utterances: each row's (utterance, frame, length) comes from an LDS table built at entry
// (two 64-way steps over p.row_starts find the tile's first utterance, a search in the 258 staged prefix sums each row's own), the
// context offsets clamp against the row's own utterance, and the epilogue scatters rows through the same table. `mtiles` carries B.
#define XS_FLAT_OFF XS_LDS_BYTES                        // rs[260] | out_row[256] | t[256] | len[256]
#define XS_FLAT_BYTES (260 * 4 + 3 * 256 * 4)

// Fused pooling on flat row tiles (flat_stats.h): the finished values replace the accumulators, then one partial sum per run of an
// utterance's rows in the wave's 128-row block. `tab`: out_row[256] | t[256] | len[256] of the tile's rows.
template <int ACT>
__device__ __forceinline__ void flat_stats_epilogue(f32x4v (&acc)[8][4], const TdnnParams& p, double* __restrict__ stats, const int* tab,
                                                    int R0, int rows_valid, int n0, int wm, int wn, int lane, const Epi16Prm& prm) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + prm.bias[j];
                if (ACT == KTF_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (ACT != KTF_ACT_NONE) v = apply_act(v, ACT);
                acc[i][j][r] = v * prm.sc[j] + prm.sh[j];
            }
    const int T = (int)p.T;
    flat_stats_runs(acc, R0, rows_valid, wm, lane,
                    [&](int m, int& t_m, int& len_m, int& b) {
                        t_m = __builtin_amdgcn_readfirstlane(tab[256 + m]);
                        len_m = __builtin_amdgcn_readfirstlane(tab[512 + m]);
                        b = (__builtin_amdgcn_readfirstlane(tab[m]) - t_m) / T;
                    },
                    [&](int b, int slot, int j, double s, double q) {
                        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
                        if (n < p.units) stats_out(stats, p, b, slot, n, s, q);
                    });
}
template <int ACT, bool STATS, bool SKIP = false, bool FLAT = false>
__global__ __launch_bounds__(512) void tdnn_x3s_kernel(TdnnParams p, int mtiles, int ntiles, int gtiles,
                                                       double* __restrict__ stats) {
    static_assert(!FLAT || !SKIP, "flat row tiling: full tiles, no row-group test");
    // LDS ring: 64 KiB stages (A hi | A lo | W hi | W lo), double buffered
    constexpr int NST = 2;
    constexpr int STG = XS_STAGE_BYTES;
    constexpr int WOFF = 2 * R_TILE_BYTES;       // W hi plane inside a stage; W lo follows it
    int fill_slot = 0, cur_slot = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = FLAT ? 0 : g / mtiles, mt = FLAT ? g : g - b * mtiles;
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
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wl + w_ob[i]), (lds_ptr_t*)(st_ + WOFF + R_TILE_BYTES + i * 8192), 16, 0, 0);
        }
    }
    // (Tried on top: the A half too, clamped to the buffer instead of the utterance -- valid while ctx[0] <= 0 -- and the
    // epilogue constants before everything: 0.7 % and 1.5 % slower.)
    if (KTF_X3_WFIRST) asm volatile("" ::: "memory");
    int len, start = 0, out_len, t0;
    const unsigned ldxb = (unsigned)p.ldx * 2u;
    int a_t[2], a_len1[2];
    unsigned a_base[2];
    if constexpr (FLAT) {
        const int B = mtiles;
        const int32_t* rs_g = p.row_starts;
        const int total = rs_g[B];
        const int R0 = mt * R_BM;
        if (R0 >= total) {
            if (KTF_X3_WFIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        int* rs = reinterpret_cast<int*>(rsm + XS_FLAT_OFF);           // prefix sums of utterances b0 .. b0 + 259
        int* trow = rs + 260;                                           // output row (b * T + t) of tile row m
        int* tt = trow + 256;                                           // its frame
        int* tlen = tt + 256;                                           // its utterance's length
        if (p.row_map) {      // the table was made once for all the step's layers (ktf_flat_row_map): one coalesced load instead of four dependent ones
            if (tid < 256) {
                const i32x4 e = reinterpret_cast<const i32x4*>(p.row_map)[R0 + tid];
                trow[tid] = e.x;
                tt[tid] = e.y;
                tlen[tid] = e.z;
            }
            __syncthreads();
        } else {
        // first utterance of the tile: the last b with row_starts[b] <= R0, by two 64-way steps (every wave, same result)
        const int S = (B + 64) >> 6;                                    // ceil((B + 1) / 64) entries per lane of the first step
        int cnt = __popcll(__ballot(lane * S <= B && rs_g[min(lane * S, B)] <= R0));
        const int k0 = (cnt - 1) * S;
        cnt = __popcll(__ballot(lane < S && k0 + lane <= B && rs_g[min(k0 + lane, B)] <= R0));
        const int b0 = __builtin_amdgcn_readfirstlane(k0 + cnt - 1);
        if (tid < 260) rs[tid] = rs_g[min(b0 + tid, B)];
        __syncthreads();
        if (tid < 256) {
            const int R = R0 + tid;
            int orow = -1, t = 0, ln = 1;
            if (R < total) {
                int lo = 0, hi = 259;                                   // last j with rs[j] <= R (empty utterances repeat a start: the last one wins)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (rs[mid] <= R) lo = mid; else hi = mid - 1;
                }
                int bb = b0 + lo;
                if (lo == 259) {                                        // (more than 259 utterance starts inside one tile: finish in global memory)
                    int l2 = bb, h2 = B - 1;
                    while (l2 < h2) {
                        const int mid = (l2 + h2 + 1) >> 1;
                        if (rs_g[mid] <= R) l2 = mid; else h2 = mid - 1;
                    }
                    bb = l2;
                }
                bb = min(bb, B - 1);
                const int s0 = rs_g[bb];
                t = R - s0;
                ln = rs_g[bb + 1] - s0;
                orow = bb * (int)p.T + t;
            }
            trow[tid] = orow;
            tt[tid] = t;
            tlen[tid] = ln;
        }
        __syncthreads();
        }
        len = (int)p.T;
        t0 = 0;
        out_len = min(R_BM, total - R0);                                // valid rows of the tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 512 + tid) >> 2;
            a_t[i] = tt[row];
            a_len1[i] = tlen[row] - 1;
            const int orow = trow[row];
            a_base[i] = orow < 0 ? 0u : (unsigned)(orow - a_t[i]) * ldxb;     // first row of the utterance, in bytes (B * T * ldx * 2 < 2^32)
        }
    } else {
        len = p.lens ? p.lens[b] : (int)p.T;
        out_len = tdnn_out_len(len, p, start);
        if (p.out_lens && nt == 0 && mt == 0 && threadIdx.x == 0) p.out_lens[b] = out_len;
        t0 = mt * R_BM;
        if (t0 >= out_len || len <= 0) {
            if (KTF_X3_WFIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing lands in the LDS of a finished workgroup
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_t[i] = start + (t0 + ((i * 512 + tid) >> 2)) * p.sub;
            a_len1[i] = len - 1;
            a_base[i] = 0;
        }
    }
    const char* xh = reinterpret_cast<const char*>(p.x) + (FLAT ? 0 : ((int64_t)b * p.T * p.ldx) * 2);      // (re-pointed by a timing ablation)
    const char* xl = reinterpret_cast<const char*>(p.x_lo) + (FLAT ? 0 : ((int64_t)b * p.T * p.ldx) * 2);

    f32x4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.ktot / R_BK;
    int is_ks = 0, is_c = 0, is_db = 0, is_off = p.ctx[0];
    const int dpad_b = p.din_pad * 2;
    // A-piece address = row * x_rm + is_xb + chunk: x_rm = row pitch, is_xb = byte offset of the 32-feature chunk in the row
    const unsigned x_rm = ldxb;
    const unsigned x_cs = (unsigned)(R_BK * 2);
    unsigned is_xb = 0;
#define XS_STAGE()                                                                                                     \
    {                                                                                                                  \
        unsigned char* st_ = rsm + fill_slot * STG + wave * 1024;                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            int r_ = a_t[i] + is_off;                                                                                  \
            r_ = r_ < 0 ? 0 : (r_ > a_len1[i] ? a_len1[i] : r_);                                                       \
            const unsigned vo_ = a_base[i] + (unsigned)r_ * x_rm + a_cb[i] + is_xb;                                    \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xh + vo_), (lds_ptr_t*)(st_ + i * 8192), 16, 0, 0);          \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xl + vo_), (lds_ptr_t*)(st_ + R_TILE_BYTES + i * 8192), 16, 0, 0); \
        }                                                                                                              \
        if (!(KTF_X3_WFIRST && is_ks == 0))                  /* stage 0's W half went out at kernel entry */           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            const unsigned vo_ = w_ob[i] + (unsigned)is_ks * w_step;                                                   \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wh + vo_), (lds_ptr_t*)(st_ + WOFF + i * 8192), 16, 0, 0);   \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wl + vo_), (lds_ptr_t*)(st_ + WOFF + R_TILE_BYTES + i * 8192), 16, 0, 0); \
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
    const Epi16Prm eprm = epi16_load(p, n0, wn, lane);      // issued here: the ~1 us of global-load latency hides under the K-loop
    const int fr = (4 - (((lane & 15) >> 2) & 3)) & 3;
    const int coff = (((lane >> 4) ^ fr) << 4);
    const int a_row_off = (wm * 128 + (lane & 15)) * 64 + coff;
    const int b_row_off = (wn * 64 + (lane & 15)) * 64 + coff;
    // 16-row blocks of this wave's 128 rows that hold a valid output row (wave-uniform)
    const int nblk = SKIP ? __builtin_amdgcn_readfirstlane(min(8, max(0, (out_len - t0 - wm * 128 + 15) >> 4))) : 8;
    {
        for (int ks = 0; ks < nk; ++ks) {
            // stage ks landed: nothing else is in flight (two stages)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            
            const unsigned char* sa = rsm + cur_slot * STG;
            const unsigned char* sw = sa + WOFF;
            cur_slot = (cur_slot + 1 == NST) ? 0 : cur_slot + 1;
            const bool refill = is_ks < nk;                     // next stage -> the buffer every wave finished reading
            unsigned char* st_ = rsm + fill_slot * STG + wave * 1024;
            // DMA n of the stage: 0,1 = A hi / lo rows 0-127; 2,3 = rows 128-255; 4,5 = W hi / lo rows 0-127; 6,7 = rows 128-255
#define XS_DMA(n)                                                                                                      \
    {                                                                                                                  \
        const char* src_ = ((n) < 4) ? ((((n) & 1) ? xl : xh) + va[((n) >> 1) & 1]) : ((((n) & 1) ? wl : wh) + vw[((n) >> 1) & 1]); \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)src_,                                                             \
            (lds_ptr_t*)(st_ + (((n) < 4) ? ((n) & 1) * R_TILE_BYTES : WOFF + ((n) & 1) * R_TILE_BYTES) + (((n) >> 1) & 1) * 8192), 16, 0, \
            0);                                                                  \
    }
            bfrag8 bh[4], bl[4], af[2][4];                      // af[set][0,1] = hi fragments of the group's two rows, [2,3] = lo
            // fragment reads in the order the MFMAs consume them (LDS returns in order: the first MFMA waits for two reads, not twelve)
            af[0][0] = *reinterpret_cast<const bfrag8*>(sa + a_row_off);
            bh[0] = *reinterpret_cast<const bfrag8*>(sw + b_row_off);
            __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise moves the A read behind the eight B reads)
#pragma unroll
            for (int j = 1; j < 4; ++j) bh[j] = *reinterpret_cast<const bfrag8*>(sw + b_row_off + j * 16 * 64);
            __builtin_amdgcn_sched_barrier(0);
            af[0][2] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off);
#pragma unroll
            for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const bfrag8*>(sw + R_TILE_BYTES + b_row_off + j * 16 * 64);
            af[0][1] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + 16 * 64);
            af[0][3] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off + 16 * 64);
            unsigned va[2], vw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int r_ = a_t[i] + is_off;
                r_ = r_ < 0 ? 0 : (r_ > a_len1[i] ? a_len1[i] : r_);
                va[i] = a_base[i] + (unsigned)r_ * x_rm + a_cb[i] + is_xb;
                vw[i] = w_ob[i] + (unsigned)is_ks * w_step;
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int PER_ROW = 12, PER_CHUNK = PER_ROW / 2;      // MFMAs per tile row / per chunk (4 chunks per 2-row group)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cur = g & 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {                  // chunk c = MFMAs 6c .. 6c+5 of the group's 24
                    if (!SKIP || 2 * g < nblk)
#pragma unroll
                    for (int m = PER_CHUNK * c; m < PER_CHUNK * c + PER_CHUNK; ++m) {
                        const int r = m / PER_ROW, j = m & 3;                  // row, column block
                        const int t = (m % PER_ROW) / 4;                        // term: 0 hh, 1 lh, 2 hl
                        f32x4v& cc = acc[2 * g + r][j];
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t == 1 ? af[cur][2 + r] : af[cur][r], t == 2 ? bl[j] : bh[j], cc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (c == 0 && g < 3) {
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            af[cur ^ 1][r] = *reinterpret_cast<const bfrag8*>(sa + a_row_off + (2 * (g + 1) + r) * 16 * 64);
                            af[cur ^ 1][2 + r] = *reinterpret_cast<const bfrag8*>(sa + R_TILE_BYTES + a_row_off + (2 * (g + 1) + r) * 16 * 64);
                        }
                    }
                    if (refill) {
                        const int n = 4 * g + c;                // slot -> DMA index
                        if (n == 0) XS_DMA(0) else if (n == 1) XS_DMA(1) else if (n == 2) XS_DMA(2) else if (n == 3) XS_DMA(3)
                        else if (n == 4) XS_DMA(4) else if (n == 5) XS_DMA(5) else if (n == 6) XS_DMA(6) else if (n == 7) XS_DMA(7)
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
    }
#undef XS_STAGE
    if (!STATS) __syncthreads();      // all fragment reads done before the LDS is reused as the store staging area
    if constexpr (STATS && FLAT)
        flat_stats_epilogue<ACT>(acc, p, stats, reinterpret_cast<const int*>(rsm + XS_FLAT_OFF) + 260, mt * R_BM, out_len, n0, wm, wn, lane, eprm);
    else
        ring_epilogue16<ACT, STATS, FLAT>(acc, p, stats, rsm, b, t0, n0, out_len, wm, wn, wave, lane, eprm,
                                          FLAT ? reinterpret_cast<const int*>(rsm + XS_FLAT_OFF) + 260 : nullptr);
}


// ------------------------------------------------------------------------------------ launcher
int tdnn_launch_split(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, int64_t ldy, bool split_in, double* stats_sums,
                      hipStream_t st) {
    {
        {
            const int mtiles = ktf_cdiv(Tout, R_BM), ntiles_r = ktf_cdiv(d->units, R_BN);
            const int64_t gtiles = B * (int64_t)mtiles;
            const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles_r;
            KTF_REQUIRE(nblocks < (1ll << 31), "ktf_tdnn: grid too large");
#define X_LAUNCH(A)                                                                                                    \
    do {                                                                                                               \
        KTF_NOTE_KERNEL("tdnn_x3r_kernel");                                                                            \
        if (stats_sums) {                                                                                              \
            KTF_LDS_ONCE(X_LDS_BYTES, tdnn_x3r_kernel<A, true>);                                                       \
            hipLaunchKernelGGL((tdnn_x3r_kernel<A, true>), dim3((unsigned)nblocks), dim3(512), X_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(X_LDS_BYTES, tdnn_x3r_kernel<A, false>);                                                      \
            hipLaunchKernelGGL((tdnn_x3r_kernel<A, false>), dim3((unsigned)nblocks), dim3(512), X_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        }                                                                                                              \
    } while (0)
            // mostly-padding tiles (fewer than 80 % of the 16-row blocks computed hold a row of a full-length utterance): the SKIP form
            const bool skip = 5 * (int64_t)ktf_cdiv(Tout, 16) < 4 * (int64_t)mtiles * 16;
#define XS_LAUNCH1(A, ST)                                                                                              \
    do {                                                                                                               \
        KTF_NOTE_KERNEL("tdnn_x3s_kernel");                                                                   \
        if (skip) {                                                                                                    \
            KTF_LDS_ONCE(XS_LDS_BYTES, tdnn_x3s_kernel<A, ST, true>);                                        \
            hipLaunchKernelGGL((tdnn_x3s_kernel<A, ST, true>), dim3((unsigned)nblocks), dim3(512), XS_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(XS_LDS_BYTES, tdnn_x3s_kernel<A, ST>);                                                        \
            hipLaunchKernelGGL((tdnn_x3s_kernel<A, ST>), dim3((unsigned)nblocks), dim3(512), XS_LDS_BYTES, st, p, mtiles, ntiles_r, (int)gtiles, stats_sums); \
        }                                                                                                              \
    } while (0)
#define XS_LAUNCH(A)                                                                                                   \
    do {                                                                                                               \
        if (stats_sums) XS_LAUNCH1(A, true); else XS_LAUNCH1(A, false);                                                \
    } while (0)
            // hi / lo planes in: the 16x16x32 plane kernel; fp32 activations in: the kernel that splits them in registers
            if (split_in && p.row_starts) {              // ktf_tdnn_split_flat: M-tiles over the batch's valid rows laid end to end
                KTF_REQUIRE(!d->valid && d->subsampling == 1, "ktf_tdnn_split_flat: SAME padding, no subsampling");
                KTF_REQUIRE(d->act == KTF_ACT_NONE || d->act == KTF_ACT_RELU, "ktf_tdnn_split_flat: fuses ReLU or no activation");
                KTF_REQUIRE(B <= 4095 && B * p.T * p.ldx * 2 < (1ll << 32), "ktf_tdnn_split_flat: B <= 4095 and B * T * ldx * 2 < 2^32");
                const int64_t ftiles = ktf_cdiv(B * p.T, R_BM);
                const int64_t fblocks = ((ftiles + 7) / 8) * 8 * ntiles_r;
                KTF_REQUIRE(fblocks < (1ll << 31), "ktf_tdnn: grid too large");
                constexpr int lds_ = XS_LDS_BYTES + XS_FLAT_BYTES;
                KTF_NOTE_KERNEL("tdnn_x3s_kernel<flat>");
                if (stats_sums) {
                    KTF_NOTE_KERNEL("tdnn_x3s_kernel<flat, pooled>");
                    if (d->act == KTF_ACT_RELU) {
                        KTF_LDS_ONCE(lds_, tdnn_x3s_kernel<KTF_ACT_RELU, true, false, true>);
                        hipLaunchKernelGGL((tdnn_x3s_kernel<KTF_ACT_RELU, true, false, true>), dim3((unsigned)fblocks), dim3(512), lds_, st, p, (int)B, ntiles_r, (int)ftiles, stats_sums);
                    } else {
                        KTF_LDS_ONCE(lds_, tdnn_x3s_kernel<KTF_ACT_NONE, true, false, true>);
                        hipLaunchKernelGGL((tdnn_x3s_kernel<KTF_ACT_NONE, true, false, true>), dim3((unsigned)fblocks), dim3(512), lds_, st, p, (int)B, ntiles_r, (int)ftiles, stats_sums);
                    }
                } else
                if (d->act == KTF_ACT_RELU) {
                    KTF_LDS_ONCE(lds_, tdnn_x3s_kernel<KTF_ACT_RELU, false, false, true>);
                    hipLaunchKernelGGL((tdnn_x3s_kernel<KTF_ACT_RELU, false, false, true>), dim3((unsigned)fblocks), dim3(512), lds_, st, p, (int)B, ntiles_r, (int)ftiles, nullptr);
                } else {
                    KTF_LDS_ONCE(lds_, tdnn_x3s_kernel<KTF_ACT_NONE, false, false, true>);
                    hipLaunchKernelGGL((tdnn_x3s_kernel<KTF_ACT_NONE, false, false, true>), dim3((unsigned)fblocks), dim3(512), lds_, st, p, (int)B, ntiles_r, (int)ftiles, nullptr);
                }
            } else
            if (split_in) {
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
        }
    }
    KTF_CHECK_LAUNCH("ktf_tdnn");
    return KTF_OK;
}
