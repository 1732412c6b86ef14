// KTF_GEMM_F16MX: TDNN layer as ONE half-precision MFMA pass plus two block-scaled (MX) residual passes at four times the
// half rate -- the gfx950-only v_mfma_scale_f32_16x16x128_f8f6f4:
//
//   y = x_h * w_h   (v_mfma_f32_16x16x32_f16;  x_h = half(x), w_h = half(w))
//     + x_l4 * w_4  (fp4 x fp4:  x_l4 = e2m1 image of the activation residual x - x_h,  w_4 = e2m1 image of w)
//     + x_4 * w_l6  (fp4 x fp6:  x_4  = e2m1 image of x_h,  w_l6 = e2m3 image of the weight residual w - w_h)
//
// every MX operand with one power-of-two scale per 32 consecutive K elements (one 32-feature chunk of one context offset).
// 1.5 MFMA passes per algorithmic flop; the x-vector deviation stays at 1.5-3.7e-5 on speech, noise and modulated noise alike
// (tools/emulate_schemes.py "mx4_46"; the half-only two-pass form is 7-10e-5 on 10 s of speech, the calibrated one-pass form
// 4-7e-4): both operands keep ~15 significant bits, and nothing depends on the input distribution. The weight residual needs
// the fp6 image (its error is the same in every frame and survives the statistics pooling); fp4 is enough for both
// activation images.
//
// Activations travel between the layers of this route as four chunk-major planes (element (b, t, d), c = d / 32):
//   xh  [b][c][t][32] half          x_h
//   xl4 [b][c][t][16 B]             32 e2m1 codes of x - x_h      (element e in nibble e: byte e/2, low nibble first)
//   x4  [b][c][t][16 B]             32 e2m1 codes of x_h
//   xs  [b][c][t] uint32            byte 0 = E8M0 scale of the xl4 block, byte 1 = of the x4 block
// written by the producing layer's epilogue (or ktf_mx_planes for the first layer), 3.125 B per activation.
//
// Kernel: 256 x 256 tile, 8 waves of 128 x 64 (8 x 4 MFMA tiles of 16 x 16), K in super-steps of four 32-deep K-steps:
//   F0..F3  one half-precision K-step each from a two-stage LDS ring (A | W images of 16 KiB, LDS-DMA);
//   M       the 64 scaled MFMAs of the super-step's 128 K elements (K-step j = K block j of the instruction) from a
//           single-buffered side area that is refilled during F0 / F1 of the NEXT super-step.
// The im2col matrix is implicit as in tdnn_gemm.hip: a K-step is one context offset of one 32-feature chunk, rows clamped
// per utterance (SAME padding = edge replication, layers/tdnn/tdnn.py:246-247 of the reference).
// Operand stream (round 6): every LDS-DMA is `buffer_load ... offen lds` through one buffer resource per tensor -- the lane's part of an
// address in a loop-invariant register, the K-step's part in a scalar register from a table of the layer's K-steps that each tile writes to
// LDS once; tiles none of whose rows is clamped (INTERIOR: three in four on 998-frame utterances, all tiles of a single-context layer) run a
// K-loop without a vector instruction for addresses; the A-row DMAs go out in the LDS-latency gap behind each K-step's barrier, every other
// DMA alone behind an MFMA group (docs/lab_notes_r6.md: what each of these is worth, and what was tried and is not here).
//
// Replaces: layers/tdnn/tdnn.py:251-280 (+ keras ReLU, batchnorm.py:78-88, stats_pooling.py:211-240 when fused).
#include "tdnn_mx_common.h"

// ------------------------------------------------------------------------------------ fp32 features -> MX planes
// One workgroup per (utterance, 32-feature chunk, 256 rows); rows at or beyond lens[b] are left unwritten (consumers clamp rows
// to len - 1). Both directions cross the LDS so that every global access is a run of consecutive bytes: the rows come in as
// 16-byte pieces in row order (one thread walking its own row touched 64 cache lines per load instruction: 0.48 ms for the
// 1024 x 998 x 30 features of a step, against 0.07 ms now), each thread then encodes one row, and the 64-byte half pieces
// leave as 16-byte pieces of consecutive records.
#define MXP_ROWS 256
__global__ __launch_bounds__(MXP_ROWS) void mx_planes_kernel(const float* __restrict__ src, int64_t T, int64_t ld, int D,
                                                             const int32_t* __restrict__ lens, char* __restrict__ xh,
                                                             char* __restrict__ xl4, char* __restrict__ x4, char* __restrict__ xs) {
    __shared__ float tile[MXP_ROWS * 33];                // row pitch 33: the row-wise reads below are conflict-free
    const int b = blockIdx.z, c = blockIdx.y, nch = gridDim.y;
    const int64_t t0 = (int64_t)blockIdx.x * MXP_ROWS;
    const int64_t len = lens ? (int64_t)lens[b] : T;
    if (t0 >= len) return;
    const int rows = (int)(len - t0 < MXP_ROWS ? len - t0 : MXP_ROWS);
    const int tid = threadIdx.x;
    const float* base = src + ((int64_t)b * T + t0) * ld + c * 32;
    const bool vec = (ld & 3) == 0 && ((uintptr_t)src & 15) == 0 && c * 32 + 32 <= ld;      // (pad columns inside the row are read and dropped)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = k * MXP_ROWS + tid, r = q >> 3, e4 = (q & 7) * 4;
        if (r < rows) {
            float v[4];
            if (vec) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(base + (int64_t)r * ld + e4);
                const int nd = D - (c * 32 + e4);          // real columns among these four
                v[0] = nd > 0 ? t.x : 0.0f; v[1] = nd > 1 ? t.y : 0.0f; v[2] = nd > 2 ? t.z : 0.0f; v[3] = nd > 3 ? t.w : 0.0f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (c * 32 + e4 + e < D) ? base[(int64_t)r * ld + e4 + e] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r * 33 + e4 + e] = __builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
        }
    }
    __syncthreads();
    const int64_t rec0 = ((int64_t)b * nch + c) * T + t0;
    u32x4 hp[4];
    if (tid < rows) {
        float v[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) v[e] = tile[tid * 33 + e];
        u32x4 l4, h4;
        unsigned sw;
        mx_encode32(v, hp, l4, h4, sw);
        *reinterpret_cast<u32x4*>(xl4 + (rec0 + tid) * 16) = l4;
        *reinterpret_cast<u32x4*>(x4 + (rec0 + tid) * 16) = h4;
        *reinterpret_cast<unsigned*>(xs + (rec0 + tid) * 4) = sw;
    }
    __syncthreads();                                     // every row has been read: the tile becomes the half-piece image
    unsigned* himg = reinterpret_cast<unsigned*>(tile);  // row pitch 20 dwords (64 B of pieces + 16 B)
    if (tid < rows) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(himg + tid * 20 + k * 4) = hp[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = k * MXP_ROWS + tid, r = q >> 2;
        if (r < rows)
            *reinterpret_cast<u32x4*>(xh + rec0 * 64 + (int64_t)q * 16) = *reinterpret_cast<const u32x4*>(himg + r * 20 + (q & 3) * 4);
    }
}

extern "C" int ktf_mx_planes(const float* src, int64_t B, int64_t T, int32_t D, int64_t ld_src, const int32_t* lens, void* xh,
                             void* xl4, void* x4, void* xs, void* stream) {
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0 && ld_src >= D, "ktf_mx_planes: bad size");
    if (B == 0 || T == 0) return KTF_OK;                 // (an empty tensor: null pointers)
    KTF_REQUIRE(src && xh && xl4 && x4 && xs, "ktf_mx_planes: null argument");
    const int nch = (D + 31) / 32;
    KTF_REQUIRE(B < 65536 && nch < 65536, "ktf_mx_planes: B and D / 32 must be below 65536");
    hipLaunchKernelGGL(mx_planes_kernel, dim3((unsigned)((T + MXP_ROWS - 1) / MXP_ROWS), (unsigned)nch, (unsigned)B), dim3(MXP_ROWS), 0,
                       (hipStream_t)stream, src, T, ld_src, (int)D, lens, (char*)xh, (char*)xl4, (char*)x4, (char*)xs);
    KTF_CHECK_LAUNCH("ktf_mx_planes");
    return KTF_OK;
}

// ------------------------------------------------------------------------------------ the GEMM
// PADK: the layer's K-steps do not fill its last super-step (nk % 4 != 0: the first layer, 5 x 32 features): the zero-padded K-steps
// skip their fragments, MFMAs and stage. An instantiation of its own: as a run-time test inside every K-step it costs the other layers 3-4 %.
// FLAT (plane output or fused pooling, SAME padding, no subsampling): the M-tiles cover the batch's VALID rows laid end to end (p.row_starts / p.row_map:
// ktf_flat_row_map) instead of 256-row tiles per utterance -- a 998-frame utterance fills 3.9 tiles, a ragged batch fewer. A tile's rows
// belong to several utterances: a thread keeps (frame, last frame, first record) of the rows it fetches, context offsets clamp against
// the row's own utterance, and the epilogue scatters rows through the same table. `mtiles` carries B. Same operands into the same MFMAs
// in the same order: the planes are bit-identical to the per-utterance tiles'.
template <int ACT, int OUT, bool PADK, bool FLAT = false>
__device__ __forceinline__ void mx_tile(const MxParams& p, const int id, int mtiles, int ntiles, int gtiles, double* __restrict__ stats,
                                        unsigned char* rsm) {
    static_assert(!FLAT || OUT != MX_OUT_F32, "flat row tiles: plane output or fused pooling");
    // Every kernel argument the tile needs before its first DMA is consumed in ONE place: the compiler otherwise loads the 240-byte parameter
    // block in five or six dependent pieces (s_load, wait, s_load, wait ...), each a trip to the scalar cache in front of the first DMA
    // issue -- a microsecond of the ~2.5 a tile spends before its first MFMA. One batch of scalar loads, one wait; placed BEHIND the tile's
    // one scalar load from global memory (the row count / the utterance's length), which it then overlaps: in front of it the asm counts as
    // a possible store and that load became a vector load.
#define MX_ARG_BATCH()                                                                                                                     \
    asm volatile("" ::"s"(p.xh), "s"(p.wh), "s"(p.T), "s"(p.nss), "s"(p.nk), "s"(p.nctx), "s"(p.nch_in), "s"(p.ctx_pk[0]), "s"(p.ctx_pk[1]), \
                 "s"(p.t_div_m), "s"(p.t_div_s), "s"(p.start), "s"(p.sub), "s"(p.cut))
    const int xcd = id & 7, slot = id >> 3;             // an XCD runs all N-tiles of an M-tile back to back (its L2 keeps the A tile)
    const int g = (slot / ntiles) * 8 + xcd;
    const int nt = slot - (slot / ntiles) * ntiles;
    if (g >= gtiles) return;
    const int b = FLAT ? 0 : g / mtiles, mt = FLAT ? g : g - b * mtiles;
    const int n0 = nt * 256, t0 = FLAT ? 0 : mt * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int R0 = mt * 256;                              // FLAT: first flat row of the tile
    int len, out_len;
    [[maybe_unused]] bool dense = false;                  // FLAT: every utterance has all T rows: row -> (utterance, frame) by division, no table load
    if constexpr (FLAT) {
        const int total = p.row_starts[mtiles];
        MX_ARG_BATCH();
        if (R0 >= total) return;
        len = (int)p.T;
        out_len = total - R0 < 256 ? total - R0 : 256;    // valid rows of the tile
        dense = total == mtiles * len;
    } else {
        len = p.lens ? p.lens[b] : (int)p.T;
        MX_ARG_BATCH();
        out_len = len - p.cut - p.start <= 0 ? 0 : (len - p.cut - p.start + p.sub - 1) / p.sub;      // (== len: SAME, no subsampling)
        if (t0 >= out_len) return;
    }
    const int lenm1 = len - 1;
    const int64_t ub = FLAT ? 0 : (int64_t)b * p.nch_in * p.T;       // first (chunk, row) record of this utterance
    const int nkp = p.nss * 4;
    // The six operand streams as buffer resources: an LDS-DMA is then `buffer_load_dwordx4 voffset, rsrc, soffset offen lds` -- the lane's
    // part of the address in ONE loop-invariant register, the K-step's part in ONE scalar register, no 64-bit address arithmetic per
    // instruction. The activation planes start MX_KQ_BIAS records early: the scalar part (chunk * T + context offset + MX_KQ_BIAS) is
    // never negative.
    const __amdgpu_buffer_rsrc_t r_xh = mx_rsrc_pinned(p.xh + (ub - MX_KQ_BIAS) * 64);
    const __amdgpu_buffer_rsrc_t r_wh = mx_rsrc_pinned(p.wh + (int64_t)nt * nkp * MX_TILE);
    const __amdgpu_buffer_rsrc_t r_wq = mx_rsrc_pinned(p.wq + (int64_t)nt * p.nss * MX_WQ_BLOCK);

    // half stage: thread q = i * 512 + tid moves the 16-byte piece at LDS offset q * 16 (row q >> 2, position q & 3 holds chunk
    // (q & 3) ^ ((4 - ((row >> 2) & 3)) & 3): conflict-free 16-byte fragment reads)
    unsigned a_cb[2];
    int a_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        const int row = q >> 2;
        a_cb[i] = (unsigned)(((q & 3) ^ ((4 - ((row >> 2) & 3)) & 3)) * 16);
        a_row[i] = p.start + (t0 + row) * p.sub;          // input row of output row t0 + row at context offset 0
    }
    // FLAT: (frame, last frame of its utterance, first record of its utterance) of the two stage rows and of the side row of this thread
    [[maybe_unused]] int a_lm1[2] = {lenm1, lenm1}, s_lm1 = lenm1;
    int s_row = p.start + (t0 + lane + 64 * (wave & 3)) * p.sub;     // side A: input row of the lane's row in the wave's row group
    [[maybe_unused]] unsigned a_ub[2] = {0u, 0u}, s_ub = 0u;
    // (output row b * T + t or -1, t, utterance length, b) of tile row m; rows beyond the batch's last: (-1, 0, 1, 0)
    [[maybe_unused]] auto flat_row = [&](int m) -> i32x4 {
        if (dense) {
            const int R = R0 + m;
            if (m >= out_len) return i32x4{-1, 0, 1, 0};
            const int bb = p.t_div_m ? (int)(__umulhi((unsigned)R, p.t_div_m) >> p.t_div_s) : R;
            return i32x4{R, R - bb * len, len, bb};
        }
        return reinterpret_cast<const i32x4*>(p.row_map)[R0 + m];
    };
    const unsigned long long cpk0 = p.ctx_pk[0], cpk1 = p.ctx_pk[1];     // the context offsets: packed signed bytes in two 64-bit kernel arguments
#define MX_CTX(ci_) ((int)(signed char)(((ci_) < 8 ? cpk0 : cpk1) >> (((ci_) & 7) * 8)))
    // INTERIOR tile (flat row tiles): all 256 rows belong to ONE utterance and no context offset leaves it -- no row of the tile is clamped in
    // any K-step (layers/tdnn/tdnn.py:246-247 of the reference replicates the edge frames: that is the clamp), so a lane's source address is
    // (its own row's record, loop-invariant) + (the K-step's chunk and context offset, a scalar): the K-loop's DMAs issue without a vector
    // instruction. 998-frame utterances on flat 256-row tiles: three tiles in four.
    [[maybe_unused]] bool interior = false;
    if constexpr (FLAT) {
        // (one branch around the three lookups: as three calls of flat_row() the table form was three dependent round trips to memory,
        // each behind its own vmcnt(0))
        i32x4 er[3];
        const int mr[3] = {tid >> 2, (512 + tid) >> 2, lane + 64 * (wave & 3)};
        if (dense) {
#pragma unroll
            for (int i = 0; i < 3; ++i) er[i] = flat_row(mr[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) er[i] = reinterpret_cast<const i32x4*>(p.row_map)[R0 + mr[i]];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_row[i] = er[i].y;
            a_lm1[i] = er[i].z - 1;
            a_ub[i] = er[i].x < 0 ? 0u : (unsigned)(er[i].x - er[i].y) * (unsigned)p.nch_in;
        }
        s_row = er[2].y;
        s_lm1 = er[2].z - 1;
        s_ub = er[2].x < 0 ? 0u : (unsigned)(er[2].x - er[2].y) * (unsigned)p.nch_in;
    }
    // The first stage goes out HERE, in front of everything only the K-loop needs (side-plane resources, the interior test, the K-step table):
    // a tile's time to its first MFMA is the latency of these four DMAs plus whatever is issued in front of them. Rows in the clamped form.
    const unsigned vw = (unsigned)tid * 16u;              // loop-invariant lane part of the W image / side W addresses
    // LDS destinations: (the wave's LDS address: wave_k, sa_k, sa_ks; inside the K-loop copies of them that pass through an empty asm once per
    // super-step) + a constant = ONE scalar add per DMA. Left to the compiler, every destination of the loop became a loop-invariant scalar
    // register of its own (~25 of them: with the six buffer resources the kernel ran out of scalar registers and spilled the epilogue's
    // pointers around the loop).
    const int wave_k0 = (int)(unsigned)(size_t)(lds_ptr_t*)(rsm + wave * 1024);       // the wave's piece of a stage image / of an 8 KiB side-W slice (LDS address)
    [[maybe_unused]] const int wave_k = wave_k0;
#define MX_LDS_AT(off_, c_) ((lds_ptr_t*)(size_t)(unsigned)((off_) + (c_)))
    int kb[5], ko[5];                                     // table entries of K-steps 4 ss .. 4 ss + 4 (scalar registers)
    kb[0] = MX_KQ_BIAS;
    ko[0] = MX_CTX(0);
    // one 16-byte-per-lane DMA of the half stage of K-step ks_ (table entry e_ of the super-step): n_ = 0, 1 the A image (rows 0-127 / 128-255);
    // _E: rows clamped to their utterance, _I: interior tiles (below)
#define MX_DMA_A_E(ks_, e_, n_)                                                                                        \
    {                                                                                                                  \
        int r_ = a_row[n_] + ko[e_];                                                                                   \
        const int hi_ = FLAT ? a_lm1[n_] : lenm1;                                                                      \
        r_ = r_ < 0 ? 0 : (r_ > hi_ ? hi_ : r_);                                                                       \
        const unsigned vo_ = ((FLAT ? a_ub[n_] : 0u) + (unsigned)r_) * 64u + a_cb[n_];                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_xh, MX_LDS_AT(wave_k, ((ks_) & 1) * MX_STAGE + (n_) * 8192), 16, vo_, kb[e_] << 6, 0, 0); \
    }
#define MX_DMA_A_I(ks_, e_, n_)                                                                                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_xh, MX_LDS_AT(wave_k, ((ks_) & 1) * MX_STAGE + (n_) * 8192), 16, (n_) ? va1 : va0, \
                                             (kb[e_] + ko[e_]) << 6, 0, 0);
#define MX_DMA_A(ks_, e_, n_) { if constexpr (INTERIOR) MX_DMA_A_I(ks_, e_, n_) else MX_DMA_A_E(ks_, e_, n_) }
    // ... n_ = 0, 1 the halves of the W image
#define MX_DMA_W(ks_, n_)                                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wh, MX_LDS_AT(wave_k, ((ks_) & 1) * MX_STAGE + MX_TILE + (n_) * 8192), 16, vw, \
                                             (ks_) * MX_TILE + (n_) * 8192, 0, 0);
    MX_DMA_W(0, 0) MX_DMA_W(0, 1) MX_DMA_A_E(0, 0, 0) MX_DMA_A_E(0, 0, 1)
    const __amdgpu_buffer_rsrc_t r_xl4 = mx_rsrc_pinned(p.xl4 + (ub - MX_KQ_BIAS) * 16);
    const __amdgpu_buffer_rsrc_t r_x4 = mx_rsrc_pinned(p.x4 + (ub - MX_KQ_BIAS) * 16);
    const __amdgpu_buffer_rsrc_t r_xs = mx_rsrc_pinned(p.xs + (ub - MX_KQ_BIAS) * 4);
    if constexpr (FLAT) {
        if (p.nctx == 1 && MX_CTX(0) == 0) {
            interior = true;      // one context at offset 0: no row is ever clamped (rows beyond the batch's last read row 0 of the plane and are not stored)
        } else if (out_len == 256) {
            const i32x4 e0 = flat_row(0), e1 = flat_row(255);
            interior = __builtin_amdgcn_readfirstlane(e0.w) == __builtin_amdgcn_readfirstlane(e1.w) &&
                       __builtin_amdgcn_readfirstlane(e0.y) + MX_CTX(0) >= 0 &&
                       __builtin_amdgcn_readfirstlane(e1.y) + MX_CTX(p.nctx - 1) <= __builtin_amdgcn_readfirstlane(e1.z) - 1;
        }
    }
    // ... and -- interior tiles -- of the lane's own rows
    [[maybe_unused]] const unsigned va0 = (a_ub[0] + (unsigned)a_row[0]) * 64u + a_cb[0], va1 = (a_ub[1] + (unsigned)a_row[1]) * 64u + a_cb[1];
    [[maybe_unused]] const unsigned vs16 = (s_ub + (unsigned)s_row) * 16u, vs4 = (s_ub + (unsigned)s_row) * 4u;

    const int sa_k0 = (int)(unsigned)(size_t)(lds_ptr_t*)(rsm + (wave & 3) * 1024 + (wave >> 2) * 4096);     // side A: the wave's pieces of the e2m1 images (+ plane, K block pair)
    const int sa_ks0 = (int)(unsigned)(size_t)(lds_ptr_t*)(rsm + (wave & 3) * 256 + (wave >> 2) * 1024);     // ... and of the scale words
    const bool wave_hi = wave >= 4;                                              // side A: the wave's K blocks are (0, 2) or (1, 3)

    // Which (32-feature chunk, context offset) a K-step reads comes from a table in LDS, made once per tile: entry k - 1 of `tkb` is
    // (chunk * T + MX_KQ_BIAS) of K-step k, of `tko` its context offset; padded K-steps (k >= nk) re-read K-step 0 (their weights are
    // zero). A super-step fetches its four entries with two 16-byte LDS reads one K-step ahead and keeps them in scalar registers: no
    // division, no scalar load (a ~200-cycle stall of the wave's whole instruction stream) and no bookkeeping between the MFMAs -- the
    // running (context index, chunk base) pairs of rounds 2-5 cost ~12 scalar instructions per half stage and ~60 per super-step's side A.
    int* const tkb = reinterpret_cast<int*>(rsm + MX_KQ_OFF);
    int* const tko = tkb + (nkp + 4);

    // side A of the super-step: n_ = 0..3 the e2m1 pieces (32 KiB: plane, K block, 64-row group by wave), 4, 5 the scale words. Piece
    // n_ * 8 + wave is (plane n_ >> 1, K block 2 (n_ & 1) + (wave >> 2), row group wave & 3): a wave fetches two of the four K blocks, the
    // same two for every piece; their table entries are chosen once per super-step by four scalar selects (sk_b / sk_o: K block
    // (wave >> 2) and 2 + (wave >> 2)), the destinations carry the wave half as a constant offset.
#define MX_DMA_SA(n_)                                                                                                  \
    {                                                                                                                  \
        constexpr int plane_ = ((n_) < 4 ? (n_) : 0) >> 1, h_ = (n_) < 4 ? ((n_) & 1) : (n_) - 4;                      \
        unsigned rec_ = 0u;                                                                                            \
        int so_;                                                                                                       \
        if constexpr (INTERIOR) {                                                                                      \
            so_ = sk_b[h_] + sk_o[h_];                                                                                 \
        } else {                                                                                                       \
            int r_ = s_row + sk_o[h_];                                                                                 \
            const int hi_ = FLAT ? s_lm1 : lenm1;                                                                      \
            r_ = r_ < 0 ? 0 : (r_ > hi_ ? hi_ : r_);                                                                   \
            rec_ = (FLAT ? s_ub : 0u) + (unsigned)r_;                                                                  \
            so_ = sk_b[h_];                                                                                            \
        }                                                                                                              \
        if ((n_) < 4) {                                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane_ ? r_x4 : r_xl4, MX_LDS_AT(sa_k, MX_SA_OFF + plane_ * 16384 + h_ * 8192), 16, \
                                                     INTERIOR ? vs16 : rec_ * 16u, so_ << 4, 0, 0);                    \
        } else {                                                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_xs, MX_LDS_AT(sa_ks, MX_SA_OFF + 32768 + h_ * 2048), 4, INTERIOR ? vs4 : rec_ * 4u, so_ << 2, 0, 0); \
        }                                                                                                              \
    }
    // side W of super-step ss_: slice n_ = 0..5 of one contiguous 48 KiB block (a wave moves its 1 KiB of every 8 KiB slice)
#define MX_DMA_SW(ss_, n_)                                                                                             \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wq, MX_LDS_AT(wave_k, MX_SW_OFF + (n_) * 8192), 16, vw, (ss_) * MX_WQ_BLOCK + (n_) * 8192, 0, 0);

    f32x4 acc[8][4];
    const int r16 = lane & 15, q4 = lane >> 4;
    const int fr = (4 - ((r16 >> 2) & 3)) & 3;
    const int coff = ((q4 ^ fr) << 4);
    const int a_row_off = (wm * 128 + r16) * 64 + coff;
    const int b_row_off = (wn * 64 + r16) * 64 + coff;
    const int klim = PADK ? p.nk : nkp;

    auto kloop = [&](auto itag) {
        constexpr bool INTERIOR = decltype(itag)::value;
        // epilogue constants of the tile's columns (bias | scale | shift -> MX_PRM_OFF): by LDS-DMA too, 64 columns per wave and vector, behind the
        // first stage (no register, no wait of their own: the K-loop's first vmcnt(0) covers them; the epilogue reads them many barriers later).
        // The resources end at the layer's last unit: columns beyond it read as 0. An absent vector is written as its neutral element.
        {
            unsigned char* const prm_w = rsm + MX_PRM_OFF + (wave & 3) * 256;
            const unsigned vcol = (unsigned)(n0 + (wave & 3) * 64 + lane) * 4u;
            const float* const vec0 = wave_hi ? p.scale : p.bias;            // waves 0-3: bias, then shift; waves 4-7: scale
            if (vec0) __builtin_amdgcn_raw_ptr_buffer_load_lds(mx_rsrc(vec0, p.units * 4), (lds_ptr_t*)(prm_w + (wave_hi ? 1024 : 0)), 4, vcol, 0, 0, 0);
            else reinterpret_cast<float*>(prm_w + (wave_hi ? 1024 : 0))[lane] = wave_hi ? 1.0f : 0.0f;
            if (!wave_hi) {
                if (p.shift) __builtin_amdgcn_raw_ptr_buffer_load_lds(mx_rsrc(p.shift, p.units * 4), (lds_ptr_t*)(prm_w + 2048), 4, vcol, 0, 0, 0);
                else reinterpret_cast<float*>(prm_w + 2048)[lane] = 0.0f;
            }
        }
        for (int k = tid; k < nkp + 4; k += 512) {        // the K-step table (the first stage is on its way); first read in F2 of super-step 0
            const int kk = k + 1 < p.nk ? k + 1 : 0;
            const int ch = kk / p.nctx, ci = kk - ch * p.nctx;
            tkb[k] = ch * (int)p.T + MX_KQ_BIAS;
            tko[k] = MX_CTX(ci);
        }
        {                                                 // K-steps 1 .. 4 directly (scalar, once per tile): no barrier in front of the loop
            int ch = 0, ci = 0;
#pragma unroll
            for (int e = 1; e <= 4; ++e) {
                if (++ci == p.nctx) { ci = 0; ++ch; }
                const bool real = e < p.nk;
                kb[e] = (real ? ch * (int)p.T : 0) + MX_KQ_BIAS;
                ko[e] = real ? MX_CTX(ci) : MX_CTX(0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int ss = 0; ss < p.nss; ++ss) {
            i32x4 tb_n, to_n;                             // the next super-step's entries: read in F2, moved to scalar registers in F3
            int sk_b[2], sk_o[2];                         // side A: the table entries of this wave's two K blocks
            // the three LDS bases of the wave's DMA destinations, passed through an empty asm ONCE per super-step: a destination is then
            // `s_add_i32 m0, base, constant` (made afresh per DMA from a loop-invariant base, every one of the ~25 destinations became a
            // loop-invariant scalar register; laundered per DMA it cost a copy and an add more each)
            int wave_k = wave_k0, sa_k = sa_k0, sa_ks = sa_ks0;
            asm volatile("" : "+s"(wave_k), "+s"(sa_k), "+s"(sa_ks));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ks = 4 * ss + j;
                // stage ks has landed; behind it only this super-step's side DMAs may still be in flight (six per wave)
                // (lgkmcnt at the head of a super-step: the K-step table and the epilogue constants written before the loop are in LDS before
                // the barrier lets anyone read them; inside the loop nothing is outstanding there)
                if (j == 1 || j == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (j == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const bool live = PADK ? ks < p.nk : true;
                // (the test is made afresh at each of its three uses: carried across the scheduling barriers as ONE boolean it went through a
                // vector register -- v_cndmask + v_cmp per K-step)
#define MX_NEXT() ({ int k1_ = ks + 1; asm volatile("" : "+s"(k1_)); k1_ < klim; })
                const unsigned char* sa = rsm + (ks & 1) * MX_STAGE;
                const unsigned char* sw = sa + MX_TILE;
                hfrag8 bh[4];
                hfrag8 a_cur, a_nx1;                      // the fragments of row blocks i and i + 1; i + 2's is read behind block i's first MFMA
                if (live) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bh[jj] = *reinterpret_cast<const hfrag8*>(sw + b_row_off + jj * 1024);
                    a_cur = *reinterpret_cast<const hfrag8*>(sa + a_row_off);
                    a_nx1 = *reinterpret_cast<const hfrag8*>(sa + a_row_off + 1024);
                }
                // the first two DMAs of the next half stage (the A rows) go out HERE, between the fragment reads above and their first use: both
                // waves of a SIMD come out of the barrier together and sit out the LDS latency in front of the first MFMA group anyway
                __builtin_amdgcn_sched_barrier(0);
                if (MX_NEXT()) { MX_DMA_A(ks + 1, j + 1, 0) MX_DMA_A(ks + 1, j + 1, 1) }
                __builtin_amdgcn_sched_barrier(0);
                // the step's other DMAs go out ONE at a time between the row blocks' MFMAs (issued in one burst behind the barrier, all eight waves
                // sit in DMA issue while the matrix pipes idle: every DMA of a K-step in that gap is 2 % slower): the W half of the next stage
                // first, then -- F0: side A, F1: side W of this super-step (the side area was released by the barrier of F0: M of ss - 1 is done)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (live) {
                        // the next row block's fragment is read BEHIND this group's first MFMA: the compiler waits for a_cur with lgkmcnt(0) right
                        // in front of the group, and with the read of a_nxt issued before that wait (as it was in every other group) the wave sat
                        // out the whole LDS latency of a fragment it needs 64 matrix cycles later -- both waves of a SIMD at the same place
                        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[0], acc[i][0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        // (TWO blocks ahead since round 6: one block of MFMAs -- 48 matrix cycles -- does not cover an LDS read's latency, and the
                        // counted lgkmcnt in front of the next block's first MFMA waited for it; with two reads in flight it waits for the older
                        // one only. Same registers in the plane kernel; the pooled layer's launch - 2.3 ... - 3 %, docs/lab_notes_r6.md 7f)
                        hfrag8 a_nx2 = a_nx1;
                        if (i < 6) a_nx2 = *reinterpret_cast<const hfrag8*>(sa + a_row_off + (i + 2) * 1024);
#pragma unroll
                        for (int jj = 1; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur, bh[jj], acc[i][jj], 0, 0, 0);
                        a_cur = a_nx1;
                        a_nx1 = a_nx2;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 0) { if (MX_NEXT()) MX_DMA_W(ks + 1, 0) }
                    if (i == 1) { if (MX_NEXT()) MX_DMA_W(ks + 1, 1) }
                    // side A (F0) / side W (F1): ONE DMA behind each of the MFMA groups 2 .. 7 (round 5 issued 2 + 2 + 1 + 1 behind groups 4 .. 7: the
                    // same instructions spread over six gaps instead of bursts in four are 1.7 % of the GEMM time, docs/lab_notes_r6.md 2a)
                    if (j == 0) {
                        if (i == 1) {
                            sk_b[0] = wave_hi ? kb[1] : kb[0]; sk_o[0] = wave_hi ? ko[1] : ko[0];
                            sk_b[1] = wave_hi ? kb[3] : kb[2]; sk_o[1] = wave_hi ? ko[3] : ko[2];
                        }
                        if (i == 2) MX_DMA_SA(0)
                        if (i == 3) MX_DMA_SA(1)
                        if (i == 4) MX_DMA_SA(2)
                        if (i == 5) MX_DMA_SA(3)
                        if (i == 6) MX_DMA_SA(4)
                        if (i == 7) MX_DMA_SA(5)
                    }
                    if (j == 1) {
                        if (i == 2) MX_DMA_SW(ss, 0)
                        if (i == 3) MX_DMA_SW(ss, 1)
                        if (i == 4) MX_DMA_SW(ss, 2)
                        if (i == 5) MX_DMA_SW(ss, 3)
                        if (i == 6) MX_DMA_SW(ss, 4)
                        if (i == 7) MX_DMA_SW(ss, 5)
                    }
                    if (j == 2 && i == 3) {               // the table entries of K-steps 4 ss + 5 .. 4 ss + 8
                        tb_n = *reinterpret_cast<const i32x4*>(tkb + 4 * ss + 4);
                        to_n = *reinterpret_cast<const i32x4*>(tko + 4 * ss + 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef MX_NEXT
            }
            // M: the two block-scaled terms of this super-step (side data complete since the barrier of F3).
            __builtin_amdgcn_sched_barrier(0);
            {
                // Fragment addresses: five byte offsets made here from the two record indices (which pass through an empty asm: the offsets
                // are then recomputed per super-step instead of living in loop-invariant registers -- that spilled), each laundered once more so
                // that every read below is `ds_read base offset:constant`: written as record arithmetic the compiler re-derived each of the
                // ~40 addresses with two or three vector instructions (55 per wave and super-step, in front of and between the scaled MFMAs).
                // are then recomputed per super-step instead of living in loop-invariant registers
                int tl;                                  // the lane index, made afresh (v_mbcnt: no register lives across the loop for it)
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tl));
                const int sw_rec = ((tl & 63) >> 4) * 256 + wn * 64 + (tl & 15);      // side W record of column block 0 (+ 16 per block)
                const int sa_rec = ((tl & 63) >> 4) * 256 + wm * 128 + (tl & 15);                      // side A record of row block 0 (+ 16 per block)
                unsigned o_a16 = MX_SA_OFF + sa_rec * 16, o_a4 = MX_SA_OFF + 32768 + sa_rec * 4;
                asm volatile("" : "+v"(o_a16), "+v"(o_a4));
                const unsigned char* const pa16 = rsm + o_a16;
                const unsigned char* const pa4 = rsm + o_a4;
                // the B fragments of all four column blocks stay resident (44 registers) while the eight row blocks stream past them
                // once: read per column half, the A side crossed the LDS twice and the M-step ran at the LDS's read rate
                // (3,008 clk of reads against 2,048 clk of MFMA per CU; now 1,856)
                unsigned o_w16 = MX_SW_OFF + sw_rec * 16, o_w8 = MX_SW_OFF + 32768 + sw_rec * 8, o_w4 = MX_SW_OFF + 40960 + sw_rec * 4;
                asm volatile("" : "+v"(o_w16), "+v"(o_w8), "+v"(o_w4));
                const unsigned char* const pw16 = rsm + o_w16;
                const unsigned char* const pw8 = rsm + o_w8;
                const unsigned char* const pw4 = rsm + o_w4;
                u32x4 w4[4], wl6a[4];
                u32x2 wl6b[4];
                unsigned wsc[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    w4[jj] = *reinterpret_cast<const u32x4*>(pw16 + jj * 256);
                    wl6a[jj] = *reinterpret_cast<const u32x4*>(pw16 + 16384 + jj * 256);
                    wl6b[jj] = *reinterpret_cast<const u32x2*>(pw8 + jj * 128);
                    wsc[jj] = *reinterpret_cast<const unsigned*>(pw4 + jj * 64);
                }
                u32x4 l_n = *reinterpret_cast<const u32x4*>(pa16);
                u32x4 h_n = *reinterpret_cast<const u32x4*>(pa16 + 16384);
                unsigned s_n = *reinterpret_cast<const unsigned*>(pa4);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const u32x4 l = l_n, h = h_n;
                    const unsigned asc = s_n;
                    if (i < 7) {                             // the next row block's fragments are read under this one's MFMAs
                        l_n = *reinterpret_cast<const u32x4*>(pa16 + (i + 1) * 256);
                        h_n = *reinterpret_cast<const u32x4*>(pa16 + 16384 + (i + 1) * 256);
                        s_n = *reinterpret_cast<const unsigned*>(pa4 + (i + 1) * 64);
                    }
                    const i32x8 al = i32x8{(int)l.x, (int)l.y, (int)l.z, (int)l.w, 0, 0, 0, 0};
                    const i32x8 ah = i32x8{(int)h.x, (int)h.y, (int)h.z, (int)h.w, 0, 0, 0, 0};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {     // residual of x (fp4, scale byte 0) times the fp4 image of w (scale byte 0)
                        const i32x8 bw = i32x8{(int)w4[jj].x, (int)w4[jj].y, (int)w4[jj].z, (int)w4[jj].w, 0, 0, 0, 0};
                        acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al, bw, acc[i][jj], 4, 4, 0, asc, 0, wsc[jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {     // fp4 image of x (scale byte 1) times the fp6 (e2m3) residual of w (scale byte 1)
                        const i32x8 bw = i32x8{(int)wl6a[jj].x, (int)wl6a[jj].y, (int)wl6a[jj].z, (int)wl6a[jj].w, (int)wl6b[jj].x, (int)wl6b[jj].y, 0, 0};
                        acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah, bw, acc[i][jj], 4, 2, 1, asc, 1, wsc[jj]);
                    }
                    if (i == 3) {                        // the next super-step's table entries become scalars (read in F2: long landed)
                        kb[0] = kb[4];
                        ko[0] = ko[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            kb[e + 1] = __builtin_amdgcn_readfirstlane(tb_n[e]);
                            ko[e + 1] = __builtin_amdgcn_readfirstlane(to_n[e]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    if constexpr (FLAT) {
        if (interior) kloop(std::true_type{});
        else kloop(std::false_type{});
    } else {
        kloop(std::false_type{});
    }
#undef MX_DMA_A
#undef MX_DMA_A_E
#undef MX_DMA_A_I
#undef MX_DMA_W
#undef MX_DMA_SA
#undef MX_DMA_SW
#undef MX_ARG_BATCH
#undef MX_LDS_AT
#undef MX_CTX

#include "tdnn_mx_epilogue.inc"
}

// (The tile body is a function of its own. A persistent form -- one workgroup per CU looping over tiles -- was measured three times and is not
// kept: round 3 for the pooled layer (67 spilled scalar registers, + 5 %), round 5 with cross-tile prefetch and a register epilogue (+ 6 %),
// round 6 as a plain loop around this body with the next tile's id from a per-XCD counter fetched at the tile's start (+ 1.8 % of GEMM time,
// same box, alternating runs: the ~1.4 us between two workgroups of a CU is not what it recovers -- docs/lab_notes_r6.md).)
template <int ACT, int OUT, bool PADK, bool FLAT = false>
__global__ __launch_bounds__(512) void tdnn_mx_kernel(MxParams p, int mtiles, int ntiles, int gtiles, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    mx_tile<ACT, OUT, PADK, FLAT>(p, blockIdx.x, mtiles, ntiles, gtiles, stats, rsm);
}

// x planes (see the head of this file) -> one TDNN layer. Exactly one of {y planes, yf, stats} is written.
static int mx_launch(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* lens,
                     const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias, const float* scale, const float* shift,
                     void* yh, void* yl4, void* y4, void* ys, float* yf, int64_t ldy, double* stats, void* stream, const char* who,
                     const int32_t* row_starts = nullptr, const int32_t* row_map = nullptr) {
    // (the descriptor and the weights are validated whether or not the input is empty; the planes of an EMPTY input -- B == 0, or T == 0
    // behind a VALID-padded layer that kept no row -- may be null pointers, as for ktf_tdnn)
    KTF_REQUIRE(d && wh && wq, "%s: null argument", who);
    KTF_REQUIRE(d->gemm == KTF_GEMM_F16MX, "%s: needs KTF_GEMM_F16MX", who);
    KTF_REQUIRE(B >= 0 && T >= 0 && B < 65536, "%s: bad size", who);
    KTF_REQUIRE(B == 0 || T == 0 || (xh && xl4 && x4 && xs), "%s: null argument", who);
    KTF_REQUIRE(d->units > 0 && d->din > 0 && d->din_pad % 32 == 0 && d->din_pad >= d->din, "%s: bad units / din / din_pad", who);
    KTF_REQUIRE(d->nctx >= 1 && d->nctx <= 16, "%s: nctx %d outside [1,16]", who, d->nctx);
    for (int i = 1; i < d->nctx; ++i) KTF_REQUIRE(d->ctx[i] > d->ctx[i - 1], "%s: context must be strictly ascending", who);
    KTF_REQUIRE(d->subsampling >= 1 && d->subsampling <= 64, "%s: subsampling %d outside [1,64]", who, d->subsampling);
    const bool plain = !d->valid && d->subsampling == 1;
    KTF_REQUIRE(plain || !stats, "%s: the fused pooling takes SAME padding without subsampling", who);
    KTF_REQUIRE(d->act == KTF_ACT_NONE || d->act == KTF_ACT_RELU, "%s: fuses ReLU or no activation", who);
    KTF_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift go together", who);
    KTF_REQUIRE(T * (int64_t)d->din_pad * 2 < (1ll << 31) - (1ll << 20), "%s: T * din_pad too large", who);
    KTF_REQUIRE((((int64_t)d->din_pad / 32 * d->nctx + 3) / 4) * 4 <= MX_KQ_MAX_STEPS || (d->flags & KTF_TDNN_MX_LOADER), "%s: more than %d K-steps (din_pad / 32 * contexts)", who, MX_KQ_MAX_STEPS);
    if (B == 0 || T == 0 || ktf_tdnn_out_len(T, d) <= 0) return KTF_OK;       // (no output row: VALID padding of an input shorter than the context)
    const int outs = (yh ? 1 : 0) + (yf ? 1 : 0) + (stats ? 1 : 0);
    KTF_REQUIRE(outs == 1, "%s: exactly one of the plane / fp32 / pooled outputs", who);
    if (yh) KTF_REQUIRE(yl4 && y4 && ys, "%s: a plane output needs all four planes", who);
    if (yf) KTF_REQUIRE(ldy >= d->units, "%s: ldy < units", who);
    if (B == 0 || T == 0) return KTF_OK;
    MxParams p;
    memset(&p, 0, sizeof(p));
    p.xh = (const char*)xh; p.xl4 = (const char*)xl4; p.x4 = (const char*)x4; p.xs = (const char*)xs;
    p.lens = lens; p.wh = (const char*)wh; p.wq = (const char*)wq; p.bias = bias; p.scale = scale; p.shift = shift;
    p.yh = (char*)yh; p.yl4 = (char*)yl4; p.y4 = (char*)y4; p.ys = (char*)ys; p.yf = yf; p.ldy = ldy; p.T = T;
    p.units = d->units; p.nch_in = d->din_pad / 32; p.nctx = d->nctx; p.nk = p.nch_in * d->nctx; p.nss = (p.nk + 3) / 4;
    p.nch_out = (d->units + 31) / 32;
    p.sub = d->subsampling;
    p.start = (d->valid && d->ctx[0] < 0) ? -d->ctx[0] : 0;
    p.cut = (d->valid && d->ctx[d->nctx - 1] > 0) ? d->ctx[d->nctx - 1] : 0;
    p.Tout = ktf_tdnn_out_len(T, d);
    if (p.Tout <= 0) return KTF_OK;
    KTF_REQUIRE(plain || !(d->flags & KTF_TDNN_MX_LOADER), "%s: the loader-wave kernel takes SAME padding without subsampling", who);
    const bool loader = (d->flags & KTF_TDNN_MX_LOADER) != 0;
    p.stat_slots = (stats && (d->flags & KTF_TDNN_DET_STATS)) ? (int32_t)ktf_mx_stats_slots(T, d->flags) : 0;
    for (int i = 0; i < d->nctx; ++i) {
        KTF_REQUIRE(d->ctx[i] >= -128 && d->ctx[i] <= 127, "%s: context offsets outside [-128, 127]", who);
        p.ctx_pk[i >> 3] |= (unsigned long long)(unsigned char)(signed char)d->ctx[i] << ((i & 7) * 8);
    }
    hipStream_t st = (hipStream_t)stream;
    const int o = stats ? MX_OUT_STATS : (yf ? MX_OUT_F32 : MX_OUT_PLANES);
    if (loader) {                               // 192 x 256 tiles, eight matrix waves + four loader waves (tdnn_mxl.hip)
        const int rc = mxl_launch(p, B, d->act, o, stats, st);
        if (rc != KTF_OK) return rc;
        KTF_CHECK_LAUNCH(who);
        return KTF_OK;
    }
    if (row_starts) {                           // flat row tiles: M-tiles over the batch's valid rows laid end to end
        KTF_REQUIRE(row_map && plain && o != MX_OUT_F32 && !loader, "%s: flat row tiles take the row table, SAME padding, no subsampling, a plane or pooled output", who);
        p.stat_slots = (stats && (d->flags & KTF_TDNN_DET_STATS)) ? (int32_t)ktf_flat_stats_slots(T) : 0;
        KTF_REQUIRE(B <= 4095 && B * T * (int64_t)p.nch_in * 64 < (1ll << 32) - (1ll << 20) && B * T < (1ll << 31) / (p.nch_out > 0 ? p.nch_out : 1),
                    "%s: flat row tiles need B <= 4095 and B * T * din_pad * 2 < 2^32", who);
        p.row_starts = row_starts; p.row_map = row_map;
        if (T > 1) {                            // x / T for x < 2^31: m = 2^(31 + l) / T + 1, s = l - 1, l = ceil(log2 T)
            uint32_t l = 0;
            while ((1ull << l) < (uint64_t)T) ++l;
            p.t_div_m = (uint32_t)((1ull << (31 + l)) / (uint64_t)T + 1);
            p.t_div_s = l - 1;
        }
        const int ntiles = ktf_cdiv(d->units, 256);
        const int64_t ftiles = ktf_cdiv(B * T, 256);
        const int64_t fblocks = ((ftiles + 7) / 8) * 8 * ntiles;
        KTF_NOTE_KERNEL("tdnn_mx_kernel<flat>");
#define MX_LAUNCH_FLAT(A, O)                                                                                           \
    {                                                                                                                  \
        if (p.nk & 3) {                                                                                                \
            KTF_LDS_ONCE(MX_LDS_TOTAL(MX_KQ_MAX_STEPS), tdnn_mx_kernel<A, O, true, true>);                                              \
            hipLaunchKernelGGL((tdnn_mx_kernel<A, O, true, true>), dim3((unsigned)fblocks), dim3(512), MX_LDS_TOTAL(p.nss * 4), st, p, (int)B, ntiles, (int)ftiles, stats); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(MX_LDS_TOTAL(MX_KQ_MAX_STEPS), tdnn_mx_kernel<A, O, false, true>);                                             \
            hipLaunchKernelGGL((tdnn_mx_kernel<A, O, false, true>), dim3((unsigned)fblocks), dim3(512), MX_LDS_TOTAL(p.nss * 4), st, p, (int)B, ntiles, (int)ftiles, stats); \
        }                                                                                                              \
    }
        if (o == MX_OUT_STATS) {
            if (d->act == KTF_ACT_RELU) MX_LAUNCH_FLAT(KTF_ACT_RELU, MX_OUT_STATS) else MX_LAUNCH_FLAT(KTF_ACT_NONE, MX_OUT_STATS)
        } else {
            if (d->act == KTF_ACT_RELU) MX_LAUNCH_FLAT(KTF_ACT_RELU, MX_OUT_PLANES) else MX_LAUNCH_FLAT(KTF_ACT_NONE, MX_OUT_PLANES)
        }
#undef MX_LAUNCH_FLAT
        KTF_CHECK_LAUNCH(who);
        return KTF_OK;
    }
    const int mtiles = ktf_cdiv(p.Tout, 256), ntiles = ktf_cdiv(d->units, 256);
    const int64_t gtiles = B * mtiles;
    const int64_t nblocks = ((gtiles + 7) / 8) * 8 * ntiles;
#define MX_LAUNCH(A, O)                                                                                                \
    {                                                                                                                  \
        KTF_NOTE_KERNEL("tdnn_mx_kernel");                                                                             \
        if (p.nk & 3) {                                                                                                \
            KTF_LDS_ONCE(MX_LDS_TOTAL(MX_KQ_MAX_STEPS), tdnn_mx_kernel<A, O, true>);                                                    \
            hipLaunchKernelGGL((tdnn_mx_kernel<A, O, true>), dim3((unsigned)nblocks), dim3(512), MX_LDS_TOTAL(p.nss * 4), st, p, mtiles, ntiles, (int)gtiles, stats); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(MX_LDS_TOTAL(MX_KQ_MAX_STEPS), tdnn_mx_kernel<A, O, false>);                                                   \
            hipLaunchKernelGGL((tdnn_mx_kernel<A, O, false>), dim3((unsigned)nblocks), dim3(512), MX_LDS_TOTAL(p.nss * 4), st, p, mtiles, ntiles, (int)gtiles, stats); \
        }                                                                                                              \
    }
    if (d->act == KTF_ACT_RELU) {
        if (o == MX_OUT_STATS) MX_LAUNCH(KTF_ACT_RELU, MX_OUT_STATS) else if (o == MX_OUT_F32) MX_LAUNCH(KTF_ACT_RELU, MX_OUT_F32) else MX_LAUNCH(KTF_ACT_RELU, MX_OUT_PLANES)
    } else {
        if (o == MX_OUT_STATS) MX_LAUNCH(KTF_ACT_NONE, MX_OUT_STATS) else if (o == MX_OUT_F32) MX_LAUNCH(KTF_ACT_NONE, MX_OUT_F32) else MX_LAUNCH(KTF_ACT_NONE, MX_OUT_PLANES)
    }
#undef MX_LAUNCH
    KTF_CHECK_LAUNCH(who);
    return KTF_OK;
}

extern "C" int ktf_tdnn_mx(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* lens,
                           const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias, const float* scale,
                           const float* shift, void* yh, void* yl4, void* y4, void* ys, float* yf, int64_t ldy, void* stream) {
    return mx_launch(xh, xl4, x4, xs, B, T, lens, d, wh, wq, bias, scale, shift, yh, yl4, y4, ys, yf, ldy, nullptr, stream, "ktf_tdnn_mx");
}

// ktf_tdnn_mx (plane output) with the M-tiles over the batch's valid rows laid end to end: bit-identical planes
extern "C" int ktf_tdnn_mx_flat(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* row_starts,
                                const int32_t* row_map, const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias, const float* scale,
                                const float* shift, void* yh, void* yl4, void* y4, void* ys, void* stream) {
    KTF_REQUIRE(B == 0 || T == 0 || (row_starts && row_map && yh), "ktf_tdnn_mx_flat: null argument");      // (empty tensors: null pointers)
    return mx_launch(xh, xl4, x4, xs, B, T, nullptr, d, wh, wq, bias, scale, shift, yh, yl4, y4, ys, nullptr, 0, nullptr, stream, "ktf_tdnn_mx_flat",
                     row_starts, row_map);
}

// ... and ktf_tdnn_mx_stats on them: one partial sum per run of an utterance's rows in a wave's 128-row block (flat_stats.h), slots and finalize
// as ktf_tdnn_split_flat_stats
extern "C" int ktf_tdnn_mx_flat_stats(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* row_starts,
                                      const int32_t* row_map, const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias,
                                      const float* scale, const float* shift, double* sums, void* stream) {
    KTF_REQUIRE(sums && (B == 0 || T == 0 || (row_starts && row_map)), "ktf_tdnn_mx_flat_stats: null argument");
    return mx_launch(xh, xl4, x4, xs, B, T, nullptr, d, wh, wq, bias, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, 0, sums, stream,
                     "ktf_tdnn_mx_flat_stats", row_starts, row_map);
}

extern "C" int ktf_tdnn_mx_stats(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T,
                                 const int32_t* lens, const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias,
                                 const float* scale, const float* shift, double* sums, void* stream) {
    KTF_REQUIRE(sums, "ktf_tdnn_mx_stats: null sums");
    return mx_launch(xh, xl4, x4, xs, B, T, lens, d, wh, wq, bias, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, 0, sums, stream,
                     "ktf_tdnn_mx_stats");
}

// KTF_TDNN_DET_STATS slots of ktf_tdnn_mx_stats: the 256-row kernel writes one slot per 128-row block (ktf_stats_slots), the loader
// kernel one per 96-row block of its 192-row tiles
extern "C" int32_t ktf_mx_slot_rows(int32_t flags) { return (flags & KTF_TDNN_MX_LOADER) ? 96 : 128; }
extern "C" int64_t ktf_mx_stats_slots(int64_t T, int32_t flags) {
    if (!(flags & KTF_TDNN_MX_LOADER)) return ktf_stats_slots(T);
    return T <= 0 ? 2 : 2 * ((T + 191) / 192);
}
