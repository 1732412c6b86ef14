// KTF_GEMM_BF16X4: fp32-grade products on the bf16 matrix pipe for FEW tiles (a single utterance, a handful of them): the small-tile
// LDS-DMA ring of tdnn_f32s_kernel (tdnn_f32.hip) with operands in the PAIR format (KTF_BF16P): every fp32 slot of an activation
// row / weight row holds bf16(v) in its low half and bf16(v - bf16(v)) in its high half -- same shapes, strides and padding as
// fp32, 16 mantissa bits.
//
// A lane's 16-byte fragment read (one chunk of its row: 4 values) IS an operand of v_mfma_f32_16x16x32_bf16: 8 bf16 slots
// (h0 l0 h1 l1 h2 l2 h3 l3). With A and B read the same way, MFMA(A, B) sums h*g + l*m over the chunk and MFMA(A', B), A' = A
// with the halves of every dword swapped (one v_alignbit each), sums l*g + h*m: all four products of (h + l)(g + m), fp32
// accumulate, no de-interleave. Two MFMAs of 16 cycles per 16 values of K against four fp32 16x16x4 MFMAs of 32 cycles, and one
// ds_read_b128 where the fp32 kernel issues four ds_read_b32 -- the two things its K-loop is bound by (tdnn_f32.hip: MFMAs + barrier
// alone 13.8 us, fragment reads + barrier alone 13.9 us, DMA stream alone 9.4 us of 22.7 us at K = 1536). What is left is the DMA stream.
//
// Replaces: layers/tdnn/tdnn.py:251-280 (+ keras ReLU, batchnorm.py:78-88) for batches too small to fill the chip on 256-row tiles,
// in every mode but "f32" (Sequential.batch_gemm): the first layer runs the fp32 kernel and writes pairs, the last one in front of
// the pooling reads pairs and writes fp32.
#include "tdnn_common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
typedef __attribute__((ext_vector_type(8))) __bf16 bpair8;

#define FS_BM 64
#define FS_NSTAGE 4
// STATS: fused StatsPooling (stats_pooling.py:231-240): the tile's fp64 column sums / sums of squares over its 64 rows go to `stats`
// (one slot per 64-row block with KTF_TDNN_DET_STATS, else atomics); the layer output is never written.
template <int BK, int BN, int NB, bool STATS>
__global__ __launch_bounds__(64 * 4 * (BN / 16 / NB)) void tdnn_x4s_kernel(TdnnParams p, double* __restrict__ stats) {
    static_assert(BN % (16 * NB) == 0, "a wave owns NB 16-column blocks");
    constexpr int WN = BN / 16 / NB;                         // waves across the tile's columns, NB blocks each
    constexpr int NT = 64 * 4 * WN;
    constexpr int CH = BK / 4;                               // 16-byte chunks per row
    constexpr int G = CH / 4;                                // MFMA pairs per K-step: lane quarter kq takes chunk 4 g + kq
    constexpr int ROWB = BK * 4;
    constexpr int A_BYTES = FS_BM * ROWB, W_BYTES = BN * ROWB;
    constexpr int TILE_BYTES = A_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr bool HALVES = (FS_BM * CH + BN * CH == NT);    // <32,64>: threads 0-511 stage A, 512-1023 stage W
    constexpr int NA = HALVES ? 1 : (FS_BM * CH) / NT;
    constexpr int NW = HALVES ? 0 : (BN * CH) / NT;
    constexpr int NDMA = HALVES ? 1 : NA + NW;
    static_assert(HALVES || ((FS_BM * CH) % NT == 0 && (BN * CH) % NT == 0), "staging does not divide");
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
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

#include "tdnn_small_ring.inc"

    // two accumulators per block -- the (A, B) and the (A', B) products -- so that consecutive MFMAs never depend on each other
    f32x4v acc[NB][2];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][h][r] = 0.0f;
    const int r16 = lane & 15, kq = lane >> 4;
    const int sw = r16 & (CH - 1);
    const int a_row_off = (wm * 16 + r16) * ROWB;
    const int b_row_off = TILE_BYTES + (wn * NB * 16 + r16) * ROWB;                 // block j: + j * 16 rows
    int frag_off[G];                                                               // chunk 4 g + kq of the lane's row
#pragma unroll
    for (int g = 0; g < G; ++g) frag_off[g] = ((4 * g + kq) ^ sw) << 4;
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
#define X4_MFMA(g_)                                                                                                    \
    {                                                                                                                  \
        const u32x4v a_ = av[g_];                                                                                      \
        const u32x4v s_ = u32x4v{__builtin_amdgcn_alignbit(a_.x, a_.x, 16), __builtin_amdgcn_alignbit(a_.y, a_.y, 16), \
                                 __builtin_amdgcn_alignbit(a_.z, a_.z, 16), __builtin_amdgcn_alignbit(a_.w, a_.w, 16)}; \
        _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                                               \
            const bpair8 b_ = __builtin_bit_cast(bpair8, bv[j][g_]);                                                   \
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bpair8, a_), b_, acc[j][0], 0, 0, 0); \
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bpair8, s_), b_, acc[j][1], 0, 0, 0); \
        }                                                                                                              \
    }
    // The fragments of step ks + 1 are read under the MFMAs of step ks; a stage is refilled four steps ahead, into the slot whose
    // fragments every wave took during the previous step (tdnn_f32s_kernel's schedule)
    u32x4v av[G], bv[NB][G];
    {
        const int issued = nk < FS_NSTAGE ? nk : FS_NSTAGE;
        FS_WAIT(issued - 1)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < G; ++g) {
            av[g] = *reinterpret_cast<const u32x4v*>(fsm + a_row_off + frag_off[g]);
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j][g] = *reinterpret_cast<const u32x4v*>(fsm + b_row_off + j * 16 * ROWB + frag_off[g]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    for (int ks = 0; ks + 1 < nk; ++ks) {
        const int beyond = nk - 2 - ks;
        FS_WAIT(beyond < 2 ? beyond : 2)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool refill = is_ks < nk;
        if (refill) {
            FS_SRC()
            FS_ADV()
        }
        const unsigned char* nst = fsm + ((ks + 1) & (FS_NSTAGE - 1)) * STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            X4_MFMA(g)
#pragma unroll
            for (int i = 0; i < NDMA; ++i)
                if ((i * G) / NDMA == g) {
                    if (refill) FS_DMA(i)
                }
            av[g] = *reinterpret_cast<const u32x4v*>(nst + a_row_off + frag_off[g]);
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j][g] = *reinterpret_cast<const u32x4v*>(nst + b_row_off + j * 16 * ROWB + frag_off[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) X4_MFMA(g)
#undef X4_MFMA
#undef FS_WAIT
#undef FS_VM
#undef FS_STAGE
#undef FS_SRC
#undef FS_DMA
#undef FS_ADV
    // 16x16 accumulator layout: acc[r] = out[row 4*(lane>>4) + r][col lane&15]
    const int rows_valid = out_len - t0;
    const int64_t out_row0 = (int64_t)b * p.Tout + t0;
    if constexpr (STATS) {
        // per column: the lane's four rows in fp64, the four lane quarters by two shuffles, the four row waves through LDS (the ring
        // is free: every wave's last fragment reads were waited for), one store / atomic pair per column
        double* red = reinterpret_cast<double*>(fsm);                    // [wm][2][BN]
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int cl = (wn * NB + j) * 16 + r16;
            const int n = n0 + cl;
            const bool nv = n < p.units;
            const float bias = (nv && p.bias) ? p.bias[n] : 0.0f;
            const float sc = (nv && p.scale) ? p.scale[n] : 1.0f;
            const float sh = (nv && p.shift) ? p.shift[n] : 0.0f;
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = wm * 16 + kq * 4 + r;
                float v = apply_act(acc[j][0][r] + acc[j][1][r] + bias, p.act);
                if (p.scale) v = v * sc + sh;
                if (m < rows_valid) {
                    s += (double)v;
                    q += (double)v * (double)v;
                }
            }
            s += __shfl_xor(s, 16, 64);
            q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (kq == 0) {
                red[(wm * 2 + 0) * BN + cl] = s;
                red[(wm * 2 + 1) * BN + cl] = q;
            }
        }
        __syncthreads();
        for (int cl = tid; cl < BN; cl += NT) {
            const int n = n0 + cl;
            if (n < p.units) {
                double s = 0.0, q = 0.0;
#pragma unroll
                for (int w_ = 0; w_ < 4; ++w_) {
                    s += red[(w_ * 2 + 0) * BN + cl];
                    q += red[(w_ * 2 + 1) * BN + cl];
                }
                stats_out(stats, p, b, blockIdx.y, n, s, q);
            }
        }
        return;
    }
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
                float v = apply_act(acc[j][0][r] + acc[j][1][r] + bias, p.act);
                if (p.scale) v = v * sc + sh;
                if (p.y_pair) v = ktf_pair(v);
                reinterpret_cast<float*>(p.y)[(out_row0 + m) * p.ldy + n] = v;
            }
        }
    }
}

// KTF_TDNN_DET_STATS slots of ktf_tdnn_stats with KTF_GEMM_BF16X4: one per 64-row tile
#define X4_SLOT_ROWS 64

int tdnn_launch_x4(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, double* stats, hipStream_t st) {
#define X4_LAUNCH(BK_, BN_, NB_)                                                                                       \
    do {                                                                                                               \
        const int lds = FS_NSTAGE * (FS_BM + BN_) * BK_ * 4;                                                           \
        dim3 grid_((unsigned)ktf_cdiv(d->units, BN_), (unsigned)ktf_cdiv(Tout, FS_BM), (unsigned)B);                   \
        KTF_NOTE_KERNEL("tdnn_x4s_kernel<" #BK_ ", " #BN_ ">");                                                         \
        if (stats) {                                                                                                   \
            KTF_LDS_ONCE(lds, tdnn_x4s_kernel<BK_, BN_, NB_, true>);                                                   \
            hipLaunchKernelGGL((tdnn_x4s_kernel<BK_, BN_, NB_, true>), grid_, dim3(64 * 4 * (BN_ / 16 / NB_)), lds, st, p, stats); \
        } else {                                                                                                       \
            KTF_LDS_ONCE(lds, tdnn_x4s_kernel<BK_, BN_, NB_, false>);                                                  \
            hipLaunchKernelGGL((tdnn_x4s_kernel<BK_, BN_, NB_, false>), grid_, dim3(64 * 4 * (BN_ / 16 / NB_)), lds, st, p, stats); \
        }                                                                                                              \
    } while (0)
    if (d->din_pad % 64 == 0) {
        // tile width as for the fp32 small tiles (tdnn_launch_f32): rounds of 256 workgroups x (width + fixed part); 96 columns
        // only where the padded W rows (the host pads to 128) cover the last tile
        const int64_t mt = (int64_t)ktf_cdiv(Tout, FS_BM) * B;
        int best = 32;
        int64_t best_cost = INT64_MAX;
        for (int bn = 32; bn <= 96; bn += 32) {
            if (bn == 96 && (int64_t)ktf_cdiv(d->units, 96) * 96 > (int64_t)ktf_cdiv(d->units, 128) * 128) continue;
            const int64_t cost = ktf_cdiv(ktf_cdiv(d->units, bn) * mt, 256) * (bn + 16);
            if (cost < best_cost) best_cost = cost, best = bn;
        }
        if (best == 32) X4_LAUNCH(64, 32, 1);
        else if (best == 64) X4_LAUNCH(64, 64, 1);
        else X4_LAUNCH(64, 96, 3);
    } else {
        X4_LAUNCH(32, 64, 1);
    }
#undef X4_LAUNCH
    KTF_CHECK_LAUNCH("ktf_tdnn");
    return KTF_OK;
}
