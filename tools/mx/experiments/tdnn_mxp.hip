// KTF_GEMM_F16MX as a PERSISTENT kernel (KTF_TDNN_MX_PERSIST): one 512-thread workgroup per CU walks its 256 x 256 tiles, and the LDS
// stage ring of tdnn_mx.hip never drains between them. Same arithmetic, same activation planes, same K-loop (four half-precision
// K-steps F0..F3 from a two-slot ring + the 64 block-scaled MFMAs M of the super-step) as tdnn_mx.hip; what changes is everything
// around the K-loop, which was 15-40 % of a tile's time there (prologue 1.3 us + plane epilogue 7.5 us + store drain, against
// 5.4 us per super-step and 2 / 4 / 12 super-steps per tile):
//
//   * the LAST K-step of a tile issues stage 0 of the workgroup's NEXT tile, so the first barrier of that tile finds its operands
//     landed: no prologue, no workgroup launch, and the plane stores of a tile drain under the next tile's K-loop;
//   * the epilogue of tile t runs INSIDE K-step 0 of tile t + 1, behind that step's barrier and behind its DMAs (stage 1 and the
//     A side of super-step 0, hoisted in front of it): the memory pipe works through 10 KiB-pieces per wave while the waves encode;
//   * the plane epilogue works from REGISTERS: the MFMA operands are swapped for the plane / fp32 outputs (weights as the A operand,
//     unit order permuted inside each 32-unit chunk of the weight images, mx.weight_images(permuted=True)), so a lane holds eight
//     consecutive units of one frame per chunk: two cross-lane maxima per 32-value block (v_permlane16/32_swap), no LDS staging
//     (the staging image was 2 us of ds_write_b32 per tile and kept the LDS busy, i.e. nothing could be prefetched under it),
//     16-byte stores of 1 KiB of consecutive records per instruction. The accumulators start at the bias.
//
// vmcnt is in order and counts stores, so what K-step 1 of the next tile waits for (stage 1) must be OLDER than the epilogue's
// stores: hence the hoisting, and a counted wait that leaves the 28 / 32 stores of a full tile in flight.
//
// LDS: as tdnn_mx.hip + a second copy of the epilogue constants (bias | scale | shift of the NEXT tile's columns, written during the
// last super-step): 157,696 B.
//
// Replaces: layers/tdnn/tdnn.py:251-280 (+ keras ReLU, batchnorm.py:78-88, stats_pooling.py:211-240 when fused).
#include "tdnn_mx_common.h"

#define XP_PRM_BYTES (3 * 256 * 4)
#define XP_LDS_BYTES (MX_PRM_OFF + 2 * XP_PRM_BYTES)
static_assert(XP_LDS_BYTES <= 163840, "LDS budget");

struct MxpParams {
    MxParams m;
    double* stats;
    int32_t mtiles, ntiles, gtiles, nids;            // M-tiles per utterance, N-tiles, B * mtiles, tile ids (gtiles rounded up to 8, x ntiles)
    uint32_t nt_m, nt_s, mt_m, mt_s;                 // x / ntiles = umulhi(x, nt_m) >> nt_s for 0 <= x < 2^31 (nt_m == 0: ntiles == 1); mtiles alike
};

// d >= 1 -> (m, s) with x / d == umulhi(x, m) >> s for every 0 <= x < 2^31; m == 0 stands for d == 1
static void xp_magic(uint32_t d, uint32_t& m, uint32_t& s) {
    if (d <= 1) { m = 0; s = 0; return; }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                     // ceil(log2 d)
    m = (uint32_t)((1ull << (31 + l)) / d + 1);
    s = l - 1;
}

typedef const __attribute__((address_space(4))) MxpParams* xp_args_t;

__device__ __forceinline__ int xp_out_len(int n, int sub) { return n <= 0 ? 0 : (sub == 1 ? n : (n + sub - 1) / sub); }

// v + (v of lane ^ 16) + (v of lane ^ 32) + (v of lane ^ 48), in every lane: the four 16-lane rows of the wave added up, by row swaps
// (no lane index, no LDS: __shfl_xor's lane id is one more register that lives through the tile loop)
__device__ __forceinline__ double xp_rows_sum(double v) {
    unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x = __longlong_as_double(((long long)b[0] << 32) | a[0]) + __longlong_as_double(((long long)b[1] << 32) | a[1]);
    lo = (unsigned)__double_as_longlong(x); hi = (unsigned)(__double_as_longlong(x) >> 32);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __longlong_as_double(((long long)b[0] << 32) | a[0]) + __longlong_as_double(((long long)b[1] << 32) | a[1]);
}

#ifdef KTF_MXP_PROF
// timing build (tools/mx/prof_mxp.py): shader-clock stamps of waves 0 and 4 (one SIMD's pair), summed per (output form, super-steps)
__device__ unsigned long long g_xp_prof[48][2][8];
#if KTF_MXP_PROF >= 2
#define XP_STAMP(k_) { if (pw >= 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[k_] += now_ - plast; plast = now_; } }
#else
#define XP_STAMP(k_)
#endif
// level 1: the K-loop of a tile (accumulator init .. last M) in shader clocks and in 100 MHz wall ticks, and the rest of the tile
#define XP_KL_BEGIN() { if (pw >= 0) { const unsigned long long c_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime(); \
                                      if (kl_c0) { pt[6] += c_ - kl_c0; pt[5] += r_ - kl_r0; } kl_c0 = c_; kl_r0 = r_; } }
#define XP_KL_END() { if (pw >= 0) { const unsigned long long c_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime(); \
                                    pt[3] += c_ - kl_c0; pt[4] += r_ - kl_r0; kl_c0 = c_; kl_r0 = r_; } }
#else
#define XP_STAMP(k_)
#define XP_KL_BEGIN()
#define XP_KL_END()
#endif

template <int ACT, int OUT, bool PADK, bool AFF>
__global__ __launch_bounds__(512) void tdnn_mxp_kernel(MxpParams q_unused) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    constexpr bool SWAP = OUT != MX_OUT_STATS;        // weights as the A operand of the MFMAs
    // The parameter block is read from the kernel-argument segment where it is needed (tile setup, epilogue) through a pointer the
    // compiler cannot see through: kept in scalar registers across the tile loop its 200 bytes spill (tdnn_mx.hip's note).
    xp_args_t kargs = (xp_args_t)__builtin_amdgcn_kernarg_segment_ptr();
#define XP_A() ({ xp_args_t a_ = kargs; asm volatile("" : "+s"(a_)); a_; })
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int stride = (int)gridDim.x;

    // ---- launch constants the K-loop uses (everything else is re-read from the argument segment where a tile is set up)
    int nk, nctx, nss, nids;
    unsigned Tu;
    unsigned long long cpk0, cpk1;
    {
        xp_args_t a = XP_A();
        Tu = (unsigned)a->m.T; nk = a->m.nk; nctx = a->m.nctx; nss = a->m.nss; nids = a->nids;
        cpk0 = a->m.ctx_pk[0]; cpk1 = a->m.ctx_pk[1];
    }
#define XP_CTX(ci_) ((int)(signed char)(((ci_) < 8 ? cpk0 : cpk1) >> (((ci_) & 7) * 8)))
    // tile id -> (utterance, first row, N-tile) with the constants a_ points at: an XCD (id & 7) runs the N-tiles of an M-tile on
    // neighbouring workgroups (its L2 keeps the rows). Divisions by multiply-high (xp_magic): everything stays in scalar registers.
#define XP_DIV(x_, m_, s_) ((m_) ? (int)(__umulhi((unsigned)(x_), (m_)) >> (s_)) : (int)(x_))
#define XP_DECODE(a_, id_, b_, t0_, nt_, ok_)                                                                          \
    {                                                                                                                  \
        const int ntiles_ = (a_)->ntiles, mtiles_ = (a_)->mtiles;                                                      \
        const int slot_ = (id_) >> 3;                                                                                  \
        const int sq_ = XP_DIV(slot_, (a_)->nt_m, (a_)->nt_s);                                                         \
        nt_ = slot_ - sq_ * ntiles_;                                                                                   \
        const int g_ = sq_ * 8 + ((id_) & 7);                                                                          \
        ok_ = (id_) < nids && g_ < (a_)->gtiles;                                                                       \
        b_ = ok_ ? XP_DIV(g_, (a_)->mt_m, (a_)->mt_s) : 0;                                                             \
        t0_ = ok_ ? (g_ - b_ * mtiles_) * 256 : 0;                                                                     \
    }
#define XP_OUT_LEN(a_, len_) xp_out_len((len_) - (a_)->m.cut - (a_)->m.start, (a_)->m.sub)

    // ---- the first tile of this workgroup. Tile state kept in scalar registers: the tile id (utterance, row and N-tile are decoded
    // from it where they are needed), len - 1 of its utterance, the operand pointers.
    int id = (int)blockIdx.x;
    int lenm1 = 0, rows_left = 0;                         // rows of the current tile's utterance at and behind the tile's first row
    {
        xp_args_t a = XP_A();
        const int32_t* lens = a->m.lens;
        for (;;) {
            if (id >= nids) return;
            bool ok;
            int b, t0, nt;
            XP_DECODE(a, id, b, t0, nt, ok)
            (void)nt;
            if (ok) {
                const int len = lens ? lens[b] : (int)Tu;
                rows_left = XP_OUT_LEN(a, len) - t0;
                lenm1 = len - 1;
                if (rows_left > 0) break;
            }
            id += stride;
        }
    }

    // ---- per-tile operand pointers and per-lane rows
    const char *xh, *xl4, *x4, *xs, *wh, *wq;
    // input row of output row t0 + r at context offset 0: t_base + r * sub. Per lane: the half stage moves rows tid >> 2 and 128 + (tid >> 2)
    // (16-byte piece tid & 3 of the row, at LDS position (tid & 3) ^ ((4 - ((row >> 2) & 3)) & 3): the same for both rows), the side A
    // pieces row `lane` of each 64-row group.
    int t_base, sub;
    // operand pointers of tile id_ (its utterance's planes, its N-tile's weight blocks); t0_ and nt_ come back for the caller
#define XP_POINTERS(a_, id_, xh_, xl4_, x4_, xs_, wh_, wq_, t0_, nt_)                                                  \
    {                                                                                                                  \
        bool ok_;                                                                                                      \
        int b_;                                                                                                        \
        XP_DECODE(a_, id_, b_, t0_, nt_, ok_)                                                                          \
        (void)ok_;                                                                                                     \
        const int64_t ub_ = (int64_t)b_ * (a_)->m.nch_in * (int64_t)Tu;                                                \
        xh_ = (a_)->m.xh + ub_ * 64;                                                                                   \
        xl4_ = (a_)->m.xl4 + ub_ * 16;                                                                                 \
        x4_ = (a_)->m.x4 + ub_ * 16;                                                                                   \
        xs_ = (a_)->m.xs + ub_ * 4;                                                                                    \
        wh_ = (a_)->m.wh + (int64_t)nt_ * (nss * 4) * MX_TILE;                                                         \
        wq_ = (a_)->m.wq + (int64_t)nt_ * nss * MX_WQ_BLOCK;                                                           \
    }
    {
        xp_args_t a = XP_A();
        int t0, nt;
        XP_POINTERS(a, id, xh, xl4, x4, xs, wh, wq, t0, nt)
        sub = a->m.sub;
        t_base = a->m.start + t0 * sub;
        // epilogue constants of the first tile's columns
        if (tid < 256) {
            float* prm = reinterpret_cast<float*>(rsm + MX_PRM_OFF);
            const int n = nt * 256 + tid;
            const bool nv = n < a->m.units;
            const float* bias = a->m.bias;
            const float* scale = a->m.scale;
            const float* shift = a->m.shift;
            prm[tid] = (nv && bias) ? bias[n] : 0.0f;
            prm[256 + tid] = (nv && scale) ? scale[n] : 1.0f;
            prm[512 + tid] = (nv && shift) ? shift[n] : 0.0f;
        }
    }

    int f_ci = 0, f_off = XP_CTX(0);                      // K-step whose half stage is issued next: context index, offset,
    unsigned f_base = 0;                                  // ... first record of its chunk (chunk * T)
    int s_ci = 0;                                         // first K-step of the super-step whose side A is issued next
    unsigned s_base = 0;
    unsigned sa_base[4];
    int sa_off[4];
#define XP_F_ADV(ks_)                                                                                                  \
    {                                                                                                                  \
        if ((ks_) < nk) {                                                                                              \
            if (++f_ci == nctx) { f_ci = 0; f_base += Tu; }                                                            \
            f_off = XP_CTX(f_ci);                                                                                      \
        } else {                                                                                                       \
            f_base = 0;                                                                                                \
            f_off = XP_CTX(0);                                                                                         \
        }                                                                                                              \
    }
    // one 16-byte-per-lane DMA of a half stage into ring slot sl_: n_ = 0, 1 the A image (rows 0-127 / 128-255) from the half plane
    // xh_ (tile base row tb_, context offset off_ of the chunk whose first record is base_, clamped to [0, lm1_]); n_ = 2, 3 the W
    // image of K-step wks_ of the N-tile's blocks wh_ (everything but tid * 16 of its address is scalar)
#define XP_DMA_F16(sl_, n_, xh_, tb_, off_, base_, lm1_, wh_, wks_)                                                    \
    {                                                                                                                  \
        unsigned char* st_ = rsm + (sl_) * MX_STAGE + wv * 1024;                                                       \
        if ((n_) < 2) {                                                                                                \
            int r_ = (tl >> 2) * sub + ((tb_) + (off_) + ((n_) & 1) * 128 * sub);                                      \
            r_ = r_ < 0 ? 0 : (r_ > (lm1_) ? (lm1_) : r_);                                                             \
            const unsigned vo_ = ((base_) + (unsigned)r_) * 64u + (unsigned)(((tl & 3) ^ ((4 - ((tl >> 4) & 3)) & 3)) * 16); \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)((xh_) + vo_), (lds_ptr_t*)(st_ + ((n_) & 1) * 8192), 16, 0, 0); \
        } else {                                                                                                       \
            const char* wb_ = (wh_) + ((unsigned)(wks_) * (unsigned)MX_TILE + (unsigned)((n_) & 1) * 8192u);           \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wb_ + (unsigned)tl * 16u), (lds_ptr_t*)(st_ + MX_TILE + ((n_) & 1) * 8192), 16, 0, 0); \
        }                                                                                                              \
    }
#define XP_DMA_CUR(ks_, n_) XP_DMA_F16((ks_) & 1, n_, xh, t_base, f_off, f_base, lenm1, wh, ks_)
    // the four K-steps of super-step ss_ (K blocks of its scaled MFMAs): chunk bases and offsets for the side A DMAs
#define XP_SA_SETUP(ss_)                                                                                               \
    {                                                                                                                  \
        _Pragma("unroll") for (int kb_ = 0; kb_ < 4; ++kb_) {                                                          \
            if (4 * (ss_) + kb_ < nk) {                                                                                \
                sa_base[kb_] = s_base;                                                                                 \
                sa_off[kb_] = XP_CTX(s_ci);                                                                            \
                if (++s_ci == nctx) { s_ci = 0; s_base += Tu; }                                                        \
            } else {                                                                                                   \
                sa_base[kb_] = 0;                                                                                      \
                sa_off[kb_] = XP_CTX(0);                                                                               \
            }                                                                                                          \
        }                                                                                                              \
    }
    // side A of the super-step set up last: n_ = 0..3 the e2m1 pieces (32 KiB: plane, K block, 64-row group by wave), 4, 5 the scale words
#define XP_DMA_SA(n_)                                                                                                  \
    {                                                                                                                  \
        const int idx_ = ((n_) < 4 ? (n_) : (n_) - 4) * 8 + wv;                                                      \
        const int plane_ = idx_ >> 4, kb_ = (n_) < 4 ? (idx_ >> 2) & 3 : idx_ >> 2, rg_ = idx_ & 3;                    \
        const unsigned base_ = kb_ == 0 ? sa_base[0] : kb_ == 1 ? sa_base[1] : kb_ == 2 ? sa_base[2] : sa_base[3];     \
        const int off__ = kb_ == 0 ? sa_off[0] : kb_ == 1 ? sa_off[1] : kb_ == 2 ? sa_off[2] : sa_off[3];              \
        int r_ = (tl & 63) * sub + (t_base + rg_ * 64 * sub + off__);                                                  \
        r_ = r_ < 0 ? 0 : (r_ > lenm1 ? lenm1 : r_);                                                                   \
        if ((n_) < 4) {                                                                                                \
            const unsigned vo_ = (base_ + (unsigned)r_) * 16u;                                                         \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)((plane_ ? x4 : xl4) + vo_),                                  \
                                             (lds_ptr_t*)(rsm + MX_SA_OFF + plane_ * 16384 + (kb_ * 256 + rg_ * 64) * 16), 16, 0, 0); \
        } else {                                                                                                       \
            const unsigned vo_ = (base_ + (unsigned)r_) * 4u;                                                          \
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(xs + vo_),                                                   \
                                             (lds_ptr_t*)(rsm + MX_SA_OFF + 32768 + (kb_ * 256 + rg_ * 64) * 4), 4, 0, 0); \
        }                                                                                                              \
    }
    // side W of super-step ss_: piece n_ = 0..5 of one contiguous 48 KiB block
#define XP_DMA_SW(ss_, n_)                                                                                             \
    {                                                                                                                  \
        const int idx_ = (n_) * 8 + wv;                                                                                \
        const unsigned vo_ = (unsigned)(ss_) * (unsigned)MX_WQ_BLOCK + (unsigned)idx_ * 1024u + (unsigned)(tl & 63) * 16u; \
        __builtin_amdgcn_global_load_lds((glb_ptr_t*)(wq + vo_), (lds_ptr_t*)(rsm + MX_SW_OFF + idx_ * 1024), 16, 0, 0); \
    }

    // (the DMA macros take the wave index and the thread index from `wv` and `tl`, copies the compiler cannot see through, made where
    // a block of DMAs starts: the wave-dependent piece addresses and select masks and the per-lane rows and offsets are then a few
    // operations there instead of ~40 scalar and ~10 vector registers that live through the whole tile loop -- and spill)
#define XP_WV() int wv = wave, tl = tid; asm volatile("" : "+s"(wv), "+v"(tl));
    // ---- stage 0 of the first tile
    { XP_WV()
    XP_DMA_CUR(0, 2) XP_DMA_CUR(0, 3) XP_DMA_CUR(0, 0) XP_DMA_CUR(0, 1) }
    XP_F_ADV(1)

    // the tile after this one: its utterance length is loaded now and looked at in the last super-step (n_id >= nids: there is none)
    int c_len;
#define XP_CANDIDATE()                                                                                                 \
    {                                                                                                                  \
        xp_args_t a_ = XP_A();                                                                                         \
        bool ok_;                                                                                                      \
        int b_, t0_, nt_;                                                                                              \
        XP_DECODE(a_, id + stride, b_, t0_, nt_, ok_)                                                                  \
        (void)t0_; (void)nt_;                                                                                          \
        const int32_t* lens_ = a_->m.lens;                                                                             \
        c_len = ok_ ? (lens_ ? lens_[b_] : (int)Tu) : 0;                                                               \
    }
    XP_CANDIDATE()
    int n_id = 0x7fffffff, n_lenm1 = 0, n_rows = 0;       // the resolved next tile (from K-step 0 of the last super-step on)
    int e_id = -1, e_rows = 0;                            // the tile whose epilogue is pending (e_id < 0: none)
    int par = 0;                                          // which copy of the epilogue constants belongs to the current tile
    bool cur_ok = true;

    const int fr = (4 - ((r16 >> 2) & 3)) & 3;
    const int coff = ((q4 ^ fr) << 4);
    const int a_row_off = (wm * 128 + r16) * 64 + coff;
    const int b_row_off = (wn * 64 + r16) * 64 + coff;

    f32x4 acc[8][4];
#ifdef KTF_MXP_PROF
    const int pw = wave == 0 ? 0 : (wave == 4 ? 1 : -1);
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_amdgcn_s_memtime(), kl_c0 = 0, kl_r0 = 0;
    (void)plast;
#endif

    for (;;) {
        // ======================================================================== opens K-step 0 of the current tile (or the drain)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        XP_STAMP(0)                                       // 0: waiting at the tile's first barrier (+ what lies between the last stamp and it)
        if (cur_ok) {
            // what K-step 0 would issue between its MFMAs goes out in front of the pending epilogue: stage 1, side A of super-step 0
            XP_WV()
            if (!PADK || 1 < nk) { XP_DMA_CUR(1, 0) XP_DMA_CUR(1, 1) XP_DMA_CUR(1, 2) XP_DMA_CUR(1, 3) }
            XP_F_ADV(2)
            XP_SA_SETUP(0)
            XP_DMA_SA(0) XP_DMA_SA(1) XP_DMA_SA(2) XP_DMA_SA(3) XP_DMA_SA(4) XP_DMA_SA(5)
        }
        __builtin_amdgcn_sched_barrier(0);
        XP_STAMP(1)                                       // 1: the hoisted DMAs
        bool e_full = false;
        if (e_id >= 0) {
            // ==================================================================== epilogue of the previous tile, from registers
            xp_args_t a = XP_A();
            // (lane and wave pass through an empty asm: what the epilogue derives from them -- store offsets, constant addresses -- is
            // then computed here, once per tile, instead of being hoisted out of the tile loop into registers that stay live through
            // the K-loop: that hoisting is what spilled when a tile loop was wrapped around tdnn_mx.hip's tile body)
            int tid_e = tid, wave_e = wave;
            asm volatile("" : "+v"(tid_e), "+s"(wave_e));
            const int r16 = tid_e & 15, q4 = (tid_e >> 4) & 3, wm = wave_e >> 2, wn = wave_e & 3;
            // (the previous tile's constants: the other copy -- or, behind the last tile, where no switch happened, this one)
            const float* prm = reinterpret_cast<const float*>(rsm + MX_PRM_OFF + (cur_ok ? par ^ 1 : par) * XP_PRM_BYTES);
            const int rows_valid = e_rows;
            int e_b, e_t0, e_n0;
            {
                bool ok_;
                int nt_;
                XP_DECODE(a, e_id, e_b, e_t0, nt_, ok_)
                (void)ok_;
                e_n0 = nt_ * 256;
            }
            if constexpr (OUT == MX_OUT_STATS) {
                // fused StatsPooling (stats_pooling.py:231-240): per unit the sum and the sum of squares of the wave's 128 rows, in fp32
                // relative to a pivot (row 0 of the block: a constant column -- a dead ReLU unit -- gives exactly 0 and 0), then fp64.
                // Accumulator (i, j)[r] = row 16 i + 4 q4 + r, image column 16 j + r16 of the wave's block.
                const int rv = rows_valid - wm * 128;
                const int units = a->m.units, stat_slots = a->m.stat_slots;
                double* stats = a->stats;
                if (rv > 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int rq = q4 * 4;                 // (opaque per column block: the 32 row masks are formed again for each, not kept
                        asm volatile("" : "+v"(rq));     // in 64 scalar registers across the four)
                        const int ul = mx_unit(wn * 4 + j, r16);
                        const float esc = prm[256 + ul], esh = prm[512 + ul];
                        const float v0 = mx_act(acc[0][j][0], ACT) * esc + esh;
                        const float pv = __int_as_float(__builtin_amdgcn_ds_bpermute(r16 << 2, __float_as_int(v0)));   // row 0's value of the column
                        float s32 = 0.0f, q32 = 0.0f;
                        int cnt = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = mx_act(acc[i][j][r], ACT) * esc + esh;
                                if (rv >= 128 || i * 16 + rq + r < rv) {
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
                        s = xp_rows_sum(s);
                        qq = xp_rows_sum(qq);
                        const int n = e_n0 + ul;
                        if (q4 == 0 && n < units) {
                            if (stat_slots > 0) {
                                double* dst = stats + (((int64_t)e_b * stat_slots + ((e_t0 >> 7) + wm)) * 2) * units + n;
                                dst[0] = s;
                                dst[units] = qq;
                            } else {
                                double* dst = stats + ((int64_t)e_b * 2) * units + n;
                                atomicAdd(dst, s);
                                atomicAdd(dst + units, qq);
                            }
                        }
                    }
                }
            } else {
                // Accumulator (i, jj)[r] = frame 16 i + r16 of the wave's rows, unit mx_unit(4 wn + jj, 4 q4 + r): for output chunk c of
                // the wave (unit blocks 2 c, 2 c + 1) this lane holds units 8 q4 .. 8 q4 + 7 of the chunk.
                const int Tout = (int)a->m.Tout;
                const int row0 = wm * 128 + r16;         // this lane's frame in row block 0 (+ 16 per block)
                if constexpr (OUT == MX_OUT_PLANES) {
                    const int nch_out = a->m.nch_out;
                    e_full = rows_valid >= 256 && (e_n0 >> 5) + 8 <= nch_out;
                    float hmax = 65504.0f;               // (the largest half, in a scalar register made here: see `zero` / `one` below)
                    asm volatile("" : "+s"(hmax));
                    // lane quarter q stores the whole e2m1 records / scale words of row block 4 g + q after a 4 x 4 transpose
                    const int rowq = wm * 128 + q4 * 16 + r16;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int chunk = (e_n0 >> 5) + wn * 2 + c;
                        if (chunk >= nch_out) continue;  // (wave-uniform)
                        // first record of the chunk's rows of this tile: uniform 64-bit bases + 32-bit per-lane offsets
                        const int64_t rec0 = ((int64_t)e_b * nch_out + chunk) * Tout + e_t0;
                        char* yh0 = a->m.yh + rec0 * 64;
                        char* yl0 = a->m.yl4 + rec0 * 16;
                        char* y40 = a->m.y4 + rec0 * 16;
                        char* ys0 = a->m.ys + rec0 * 4;
                        float esc8[8], esh8[8];
                        if constexpr (AFF) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int ul = mx_unit(wn * 4 + 2 * c + h, q4 * 4);
                                const f32x4 s4 = *reinterpret_cast<const f32x4*>(prm + 256 + ul);
                                const f32x4 h4 = *reinterpret_cast<const f32x4*>(prm + 512 + ul);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { esc8[4 * h + e] = s4[e]; esh8[4 * h + e] = h4[e]; }
                            }
                        }
#pragma unroll
                        for (int g4 = 0; g4 < 2; ++g4) {
                            unsigned l4r[4], h4r[4], swr[4];
#pragma unroll
                            for (int i2 = 0; i2 < 4; i2 += 2) {
                                float v[2][8];
#pragma unroll
                                for (int n = 0; n < 2; ++n)
#pragma unroll
                                    for (int e = 0; e < 8; ++e) {
                                        const float x = acc[4 * g4 + i2 + n][2 * c + (e >> 2)][e & 3];
                                        // the planes saturate at the largest half: one v_med3 does the ReLU and the clamp
                                        if constexpr (AFF) v[n][e] = __builtin_amdgcn_fmed3f(mx_act(x, ACT) * esc8[e] + esh8[e], -hmax, hmax);
                                        else v[n][e] = __builtin_amdgcn_fmed3f(x, ACT == KTF_ACT_RELU ? 0.0f : -hmax, hmax);
                                    }
                                u32x4 hp[2];
                                unsigned l4p[2], h4p[2], swp[2];
                                mx_encode8<2>(v, hp, l4p, h4p, swp);
#pragma unroll
                                for (int n = 0; n < 2; ++n) {
                                    const int ib = i2 + n;
                                    l4r[ib] = l4p[n]; h4r[ib] = h4p[n]; swr[ib] = swp[n];
                                    unsigned rr = (unsigned)(row0 + (4 * g4 + ib) * 16);
                                    asm volatile("" : "+v"(rr));      // (the address is formed here, not hoisted and kept in registers)
                                    if ((int)rr < rows_valid)
                                        __builtin_nontemporal_store(hp[n], reinterpret_cast<u32x4*>(yh0 + (rr * 64u + (unsigned)q4 * 16u)));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            mx_transpose4(l4r[0], l4r[1], l4r[2], l4r[3]);
                            mx_transpose4(h4r[0], h4r[1], h4r[2], h4r[3]);
                            unsigned rq = (unsigned)(rowq + g4 * 64);
                            asm volatile("" : "+v"(rq));
                            if ((int)rq < rows_valid) {
                                const unsigned sw = q4 == 0 ? swr[0] : q4 == 1 ? swr[1] : q4 == 2 ? swr[2] : swr[3];
                                __builtin_nontemporal_store(u32x4{l4r[0], l4r[1], l4r[2], l4r[3]}, reinterpret_cast<u32x4*>(yl0 + rq * 16u));
                                __builtin_nontemporal_store(u32x4{h4r[0], h4r[1], h4r[2], h4r[3]}, reinterpret_cast<u32x4*>(y40 + rq * 16u));
                                __builtin_nontemporal_store(sw, reinterpret_cast<unsigned*>(ys0 + rq * 4u));
                            }
                        }
                    }
                } else {
                    // fp32 rows (B, Tout, ldy): four consecutive units per accumulator
                    const int64_t ldy = a->m.ldy;
                    const int units = a->m.units;
                    float* yf = a->m.yf + ((int64_t)e_b * Tout + e_t0) * ldy;
                    const bool vec = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a->m.yf) & 15) == 0;
                    e_full = rows_valid >= 256 && e_n0 + 256 <= units && vec;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int ul = mx_unit(wn * 4 + jj, q4 * 4);
                        const int n = e_n0 + ul;
                        f32x4 es = f32x4{1.0f, 1.0f, 1.0f, 1.0f}, eh = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        if constexpr (AFF) {
                            es = *reinterpret_cast<const f32x4*>(prm + 256 + ul);
                            eh = *reinterpret_cast<const f32x4*>(prm + 512 + ul);
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int row = row0 + i * 16;
                            if (row >= rows_valid) continue;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = mx_act(acc[i][jj][e], ACT);
                                if constexpr (AFF) v[e] = v[e] * es[e] + eh[e];
                            }
                            float* yp = yf + (int64_t)row * ldy + n;
                            if (vec && n + 4 <= units) {
                                *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (n + e < units) yp[e] = v[e];
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        XP_STAMP(2)                                       // 2: the epilogue
#ifdef KTF_MXP_PROF
        if (pw >= 0 && e_id >= 0) ++pt[7];               // (tiles; the sums leave once, behind the tile loop: atomics inside it would sit in
                                                          // the in-order vmcnt queue in front of the next tile's stages)
#endif
        if (!cur_ok) break;

        XP_KL_BEGIN()
        // the accumulators start at the bias of their units
        {
            int tid_e = tid, wave_e = wave;
            asm volatile("" : "+v"(tid_e), "+s"(wave_e));
            const int r16 = tid_e & 15, q4 = (tid_e >> 4) & 3, wn = wave_e & 3;
            const float* prm = reinterpret_cast<const float*>(rsm + MX_PRM_OFF + par * XP_PRM_BYTES);
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
                for (int i = 0; i < 8; ++i) acc[i][j] = b4;
            }
        }

        for (int ss = 0; ss < nss; ++ss) {
            const bool last_ss = ss + 1 == nss;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ks = 4 * ss + j;
                if (j != 0 || ss != 0) {
                    // stage ks has landed; behind it only this super-step's side DMAs may still be in flight (six per wave) -- and, in
                    // K-step 1 of a tile, the stores of the epilogue that ran in K-step 0 (28 plane / 32 fp32 stores of a full tile)
                    if (j == 1) {
                        if (ss == 0 && e_full) {
                            if constexpr (OUT == MX_OUT_PLANES) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        }
                    } else if (j == 2) {
                        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                XP_WV()
                if (j == 0 && last_ss) {
                    // the next tile of this workgroup: the candidate, or -- its M-tile is empty (ragged batch) -- the next one that is not
                    xp_args_t a = XP_A();
                    const int32_t* lens = a->m.lens;
                    n_id = id + stride;
                    int n_nt = 0;
                    (void)n_nt;
                    for (int n_len = c_len;; ) {
                        bool ok;
                        int n_b, n_t0;
                        XP_DECODE(a, n_id, n_b, n_t0, n_nt, ok)
                        if (n_id >= nids) break;
                        if (ok) {
                            n_rows = XP_OUT_LEN(a, n_len) - n_t0;
                            n_lenm1 = n_len - 1;
                            if (n_rows > 0) break;
                        }
                        n_id += stride;
                        XP_DECODE(a, n_id, n_b, n_t0, n_nt, ok)
                        n_len = ok ? (lens ? lens[n_b] : (int)Tu) : 0;
                    }
                }
                if (j == 1 && last_ss && n_id < nids && tid < 256) {
                    // the next tile's column constants go into the other LDS copy (whose last reader, the epilogue that ran in K-step 0
                    // of this tile, is behind this K-step's barrier): by LDS-DMA where the unit (and the array) exists -- lanes that
                    // are switched off write nothing -- and as the default elsewhere. No register carries them across a K-step, and
                    // they are older than this K-step's stage: the next barrier's wait covers them.
                    xp_args_t a = XP_A();
                    bool ok_;
                    int b_, t0_, nt_;
                    XP_DECODE(a, n_id, b_, t0_, nt_, ok_)
                    (void)ok_; (void)b_; (void)t0_;
                    unsigned char* prm = rsm + MX_PRM_OFF + (par ^ 1) * XP_PRM_BYTES + wv * 256;
                    const int n = nt_ * 256 + tl;
                    const bool nv = n < a->m.units;
                    const float* bias = a->m.bias;
                    const float* scale = a->m.scale;
                    const float* shift = a->m.shift;
                    float zero = 0.0f, one = 1.0f;       // (made here: as plain constants they sit in two registers through the whole tile loop)
                    asm volatile("" : "+v"(zero), "+v"(one));
                    if (nv && bias) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(bias + n), (lds_ptr_t*)prm, 4, 0, 0);
                    else reinterpret_cast<float*>(prm)[tl & 63] = zero;
                    if (nv && scale) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(scale + n), (lds_ptr_t*)(prm + 1024), 4, 0, 0);
                    else reinterpret_cast<float*>(prm + 1024)[tl & 63] = one;
                    if (nv && shift) __builtin_amdgcn_global_load_lds((glb_ptr_t*)(shift + n), (lds_ptr_t*)(prm + 2048), 4, 0, 0);
                    else reinterpret_cast<float*>(prm + 2048)[tl & 63] = zero;
                }
                const bool live = PADK ? ks < nk : true;
                const bool step0 = j == 0 && ss == 0;                        // its DMAs went out in front of the epilogue
                const bool next = PADK ? ks + 1 < nk : !(j == 3 && last_ss);
                const bool cross = j == 3 && last_ss && n_id < nids;        // this K-step issues stage 0 of the NEXT tile
                const unsigned char* sa = rsm + (ks & 1) * MX_STAGE;
                const unsigned char* sw = sa + MX_TILE;
                hfrag8 bh[4];
                hfrag8 a_cur;
                if (live) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bh[jj] = *reinterpret_cast<const hfrag8*>(sw + b_row_off + jj * 1024);
                    a_cur = *reinterpret_cast<const hfrag8*>(sa + a_row_off);
                }
                // next tile's operands (K-step 3 of the last super-step)
                const char* xh_n = xh;
                const char* wh_n = wh;
                int tb_n = 0, lm1_n = 0;
                if (j == 3 && cross) {
                    xp_args_t a = XP_A();
                    bool ok_;
                    int b_, t0_, nt_;
                    XP_DECODE(a, n_id, b_, t0_, nt_, ok_)
                    (void)ok_;
                    xh_n = a->m.xh + (int64_t)b_ * a->m.nch_in * (int64_t)Tu * 64;
                    wh_n = a->m.wh + (int64_t)nt_ * (nss * 4) * MX_TILE;
                    tb_n = a->m.start + t0_ * sub;
                    lm1_n = n_lenm1;
                }
                const int off0 = XP_CTX(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (live) {
                        hfrag8 a_nxt = a_cur;
                        if (i < 7) a_nxt = *reinterpret_cast<const hfrag8*>(sa + a_row_off + (i + 1) * 1024);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            acc[i][jj] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[jj], a_cur, acc[i][jj], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);
                        a_cur = a_nxt;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!step0) {
                        if (next && !(j == 3 && last_ss)) {
                            if (i == 0) XP_DMA_CUR(ks + 1, 0)
                            if (i == 1) { XP_DMA_CUR(ks + 1, 1) XP_F_ADV(ks + 2) }
                            if (i == 2) XP_DMA_CUR(ks + 1, 2)
                            if (i == 3) XP_DMA_CUR(ks + 1, 3)
                        }
                        if (j == 3 && cross) {
                            if (i == 0) XP_DMA_F16(0, 0, xh_n, tb_n, off0, 0u, lm1_n, wh_n, 0)
                            if (i == 1) XP_DMA_F16(0, 1, xh_n, tb_n, off0, 0u, lm1_n, wh_n, 0)
                            if (i == 2) XP_DMA_F16(0, 2, xh_n, tb_n, off0, 0u, lm1_n, wh_n, 0)
                            if (i == 3) XP_DMA_F16(0, 3, xh_n, tb_n, off0, 0u, lm1_n, wh_n, 0)
                        }
                        if (j == 0) {
                            if (i == 3) XP_SA_SETUP(ss)
                            if (i == 4) { XP_DMA_SA(0) XP_DMA_SA(1) }
                            if (i == 5) { XP_DMA_SA(2) XP_DMA_SA(3) }
                            if (i == 6) XP_DMA_SA(4)
                            if (i == 7) XP_DMA_SA(5)
                        }
                    }
                    if (j == 1) {
                        if (i == 4) { XP_DMA_SW(ss, 0) XP_DMA_SW(ss, 1) }
                        if (i == 5) { XP_DMA_SW(ss, 2) XP_DMA_SW(ss, 3) }
                        if (i == 6) XP_DMA_SW(ss, 4)
                        if (i == 7) XP_DMA_SW(ss, 5)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // M: the two block-scaled terms of this super-step (side data complete since the barrier of F3).
            __builtin_amdgcn_sched_barrier(0);
            XP_STAMP(ss == 0 ? 3 : 4)                     // 3: F0-F3 of super-step 0 (with the accumulator init), 4: of the others
            {
                // (the two record indices pass through an empty asm: the fragment addresses are then recomputed here, a few VALU
                // operations per super-step, instead of living in ~20 loop-invariant registers -- which is what spilled)
                int tm = tid, wvm = wave;
                asm volatile("" : "+v"(tm), "+s"(wvm));
                const int rq_ = ((tm >> 4) & 3) * 256 + (tm & 15);
                const int sw_rec = rq_ + (wvm & 3) * 64;                 // side W record of column block 0 (+ 16 per block)
                const int sa_rec = rq_ + (wvm >> 2) * 128;               // side A record of row block 0
                const unsigned char* sA = rsm + MX_SA_OFF;
                const unsigned char* sW = rsm + MX_SW_OFF;
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
                u32x4 l_n = *reinterpret_cast<const u32x4*>(sA + sa_rec * 16);
                u32x4 h_n = *reinterpret_cast<const u32x4*>(sA + 16384 + sa_rec * 16);
                unsigned s_n = *reinterpret_cast<const unsigned*>(sA + 32768 + sa_rec * 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u32x4 l = l_n, h = h_n;
                    const unsigned asc = s_n;
                    if (i < 7) {                         // the next row block's fragments are read under this one's MFMAs
                        const int rec = sa_rec + (i + 1) * 16;
                        l_n = *reinterpret_cast<const u32x4*>(sA + rec * 16);
                        h_n = *reinterpret_cast<const u32x4*>(sA + 16384 + rec * 16);
                        s_n = *reinterpret_cast<const unsigned*>(sA + 32768 + rec * 4);
                    }
                    const i32x8 al = i32x8{(int)l.x, (int)l.y, (int)l.z, (int)l.w, 0, 0, 0, 0};
                    const i32x8 ah = i32x8{(int)h.x, (int)h.y, (int)h.z, (int)h.w, 0, 0, 0, 0};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {     // residual of x (fp4, scale byte 0) times the fp4 image of w (scale byte 0)
                        const i32x8 bw = i32x8{(int)w4[jj].x, (int)w4[jj].y, (int)w4[jj].z, (int)w4[jj].w, 0, 0, 0, 0};
                        acc[i][jj] = SWAP ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, al, acc[i][jj], 4, 4, 0, wsc[jj], 0, asc)
                                          : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al, bw, acc[i][jj], 4, 4, 0, asc, 0, wsc[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {     // fp4 image of x (scale byte 1) times the fp6 (e2m3) residual of w (scale byte 1)
                        const i32x8 bw = i32x8{(int)wl6a[jj].x, (int)wl6a[jj].y, (int)wl6a[jj].z, (int)wl6a[jj].w, (int)wl6b[jj].x, (int)wl6b[jj].y, 0, 0};
                        acc[i][jj] = SWAP ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bw, ah, acc[i][jj], 2, 4, 1, wsc[jj], 1, asc)
                                          : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah, bw, acc[i][jj], 4, 2, 1, asc, 1, wsc[jj]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            XP_STAMP(5)                                   // 5: the M steps
        }
        XP_KL_END()
        // ======================================================================== on to the next tile (its stage 0 is in flight)
        e_id = id; e_rows = rows_left;
        cur_ok = n_id < nids;
        if (cur_ok) {
            xp_args_t a = XP_A();
            int t0_new, nt_;
            id = n_id;
            XP_POINTERS(a, id, xh, xl4, x4, xs, wh, wq, t0_new, nt_)
            t_base = a->m.start + t0_new * sub;
            lenm1 = n_lenm1; rows_left = n_rows;
            par ^= 1;
            f_ci = 0; f_base = 0; f_off = XP_CTX(0);
            XP_F_ADV(1)
            s_ci = 0; s_base = 0;
            XP_CANDIDATE()
        }
        XP_STAMP(6)                                       // 6: the switch to the next tile
    }
#ifdef KTF_MXP_PROF
    if (pw >= 0 && (tid & 63) == 0) {
        unsigned long long* g = g_xp_prof[(nss < 15 ? nss : 15) + 16 * OUT][pw];
        for (int k = 0; k < 8; ++k) atomicAdd(g + k, pt[k]);
    }
#endif
#undef XP_A
#undef XP_CTX
#undef XP_DECODE
#undef XP_OUT_LEN
#undef XP_POINTERS
#undef XP_DIV
#undef XP_F_ADV
#undef XP_DMA_F16
#undef XP_DMA_CUR
#undef XP_SA_SETUP
#undef XP_DMA_SA
#undef XP_DMA_SW
#undef XP_CANDIDATE
#undef XP_WV
}

// tdnn_mx.hip's launcher hands over `p` (filled from the descriptor) when KtfTdnnDesc.flags has KTF_TDNN_MX_PERSIST
int mxp_launch(const MxParams& p, int64_t B, int act, int out_kind, double* stats, hipStream_t st) {
    MxpParams q;
    memset(&q, 0, sizeof(q));
    q.m = p;
    q.stats = stats;
    q.mtiles = ktf_cdiv(p.Tout, 256);
    q.ntiles = ktf_cdiv(p.units, 256);
    const int64_t gtiles = B * q.mtiles;
    const int64_t nids = ((gtiles + 7) / 8) * 8 * q.ntiles;
    KTF_REQUIRE(nids < (1ll << 31) - 65536, "ktf_tdnn_mx: too many tiles for the persistent kernel");
    q.gtiles = (int32_t)gtiles;
    q.nids = (int32_t)nids;
    xp_magic((uint32_t)q.ntiles, q.nt_m, q.nt_s);
    xp_magic((uint32_t)q.mtiles, q.mt_m, q.mt_s);
    // one workgroup per CU (its 157,696 B of LDS admit no second one); a multiple of 8 so that a workgroup stays on one XCD's tile ids
    static int cus[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int ncu = cus[dev & 63];
    if (ncu == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        ncu = (ncu / 8) * 8;
        if (ncu < 8) ncu = 8;
        cus[dev & 63] = ncu;
    }
    const int64_t grid = nids < ncu ? nids : ncu;
    const bool aff = p.scale != nullptr;
#define XP_LAUNCH4(A, O, PK, AF)                                                                                       \
    {                                                                                                                  \
        KTF_LDS_ONCE(XP_LDS_BYTES, tdnn_mxp_kernel<A, O, PK, AF>);                                                     \
        hipLaunchKernelGGL((tdnn_mxp_kernel<A, O, PK, AF>), dim3((unsigned)grid), dim3(512), XP_LDS_BYTES, st, q);     \
    }
    // (the pooled form applies scale / shift from the LDS constants whether they were given or not: one instantiation)
#define XP_LAUNCH_POOL(A) { if (p.nk & 3) XP_LAUNCH4(A, MX_OUT_STATS, true, false) else XP_LAUNCH4(A, MX_OUT_STATS, false, false) }
#define XP_LAUNCH_ROWS(A, O)                                                                                           \
    {                                                                                                                  \
        if (aff) {                                                                                                     \
            if (p.nk & 3) XP_LAUNCH4(A, O, true, true) else XP_LAUNCH4(A, O, false, true)                              \
        } else {                                                                                                       \
            if (p.nk & 3) XP_LAUNCH4(A, O, true, false) else XP_LAUNCH4(A, O, false, false)                            \
        }                                                                                                              \
    }
    KTF_NOTE_KERNEL("tdnn_mxp_kernel");
    if (act == KTF_ACT_RELU) {
        if (out_kind == MX_OUT_STATS) XP_LAUNCH_POOL(KTF_ACT_RELU) else if (out_kind == MX_OUT_F32) XP_LAUNCH_ROWS(KTF_ACT_RELU, MX_OUT_F32) else XP_LAUNCH_ROWS(KTF_ACT_RELU, MX_OUT_PLANES)
    } else {
        if (out_kind == MX_OUT_STATS) XP_LAUNCH_POOL(KTF_ACT_NONE) else if (out_kind == MX_OUT_F32) XP_LAUNCH_ROWS(KTF_ACT_NONE, MX_OUT_F32) else XP_LAUNCH_ROWS(KTF_ACT_NONE, MX_OUT_PLANES)
    }
#undef XP_LAUNCH_ROWS
#undef XP_LAUNCH_POOL
#undef XP_LAUNCH4
    return KTF_OK;
}

#ifdef KTF_MXP_PROF
extern "C" void ktf_prof_dump(void) {
    unsigned long long h[48][2][8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_xp_prof), sizeof(h));
    for (int i = 0; i < 48; ++i)
        for (int w = 0; w < 2; ++w)
            if (h[i][w][7]) {
                const double n = (double)h[i][w][7];
#if KTF_MXP_PROF >= 2
                printf("out %d nss %2d wave %d: %llu tiles | clk per tile: first barrier %.0f  hoisted DMAs %.0f  epilogue %.0f  init + F(ss 0) %.0f  F(ss > 0) %.0f  M %.0f  switch %.0f\n",
                       i / 16, i % 16, w * 4, h[i][w][7], h[i][w][0] / n, h[i][w][1] / n, h[i][w][2] / n, h[i][w][3] / n, h[i][w][4] / n, h[i][w][5] / n, h[i][w][6] / n);
#else
                printf("out %d nss %2d wave %d: %llu tiles | K-loop %.0f clk = %.2f us (%.0f MHz) | rest of the tile %.0f clk = %.2f us\n",
                       i / 16, i % 16, w * 4, h[i][w][7], h[i][w][3] / n, h[i][w][4] / n / 100.0, 100.0 * (double)h[i][w][3] / (double)h[i][w][4],
                       h[i][w][6] / n, h[i][w][5] / n / 100.0);
#endif
            }
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xp_prof), h, sizeof(h));
}
#endif
