// KTF_GEMM_F32: exact fp32 products / fp32 accumulate on v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (bit-identical to an fmaf chain in K
// order): the parity path. Throughput tile (128 x 128, two workgroups per CU), single-utterance tiles (64 x 32 / 64 / 96), a
// row-vector kernel for <= 8 output rows, and the register-staged 32x32x2 kernels kept as their bitwise reference
// (KTF_TDNN_REF_TILES).
#include "tdnn_common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4v;

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
        __builtin_amdgcn_s_barrier();                         // every wave has taken stage ks (its reads were waited for at the
        asm volatile("" ::: "memory");                       // end of the previous step): the slot can be refilled
        const bool refill = is_ks < nk;
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
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[j][c], acc[j], 0, 0, 0);
            }
            if (c % DSTEP == 0 && c / DSTEP < NDMA) {
                if (refill) FS_DMA(c / DSTEP)
            }
            av[c] = *reinterpret_cast<const float*>(nst + a_row_off + ((c ^ sw) << 4));
#pragma unroll
            for (int j = 0; j < NB; ++j)
                bv[j][c] = *reinterpret_cast<const float*>(nst + b_row_off + j * 16 * ROWB + ((c ^ sw) << 4));
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
                if (p.y_pair) v = ktf_pair(v);
                const int64_t off = (out_row0 + m) * p.ldy + n;
                if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off] = v;
                else reinterpret_cast<unsigned short*>(p.y)[off] = f2bf(v);
            }
        }
    }
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
                if (p.y_pair) v[e] = ktf_pair(v[e]);
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
        if (p.y_pair) v = ktf_pair(v);
        const int64_t off = ((int64_t)b * p.Tout + t) * p.ldy + n;
        if (p.y_dtype == KTF_F32) reinterpret_cast<float*>(p.y)[off] = v;
        else reinterpret_cast<unsigned short*>(p.y)[off] = f2bf(v);
    }
}


// ------------------------------------------------------------------------------------ launcher
int tdnn_launch_f32(const TdnnParams& p, const KtfTdnnDesc* d, int64_t B, int64_t Tout, hipStream_t st) {
        KTF_REQUIRE(d->x_dtype == KTF_F32 && d->w_dtype == KTF_F32, "ktf_tdnn: F32 gemm needs fp32 x and w");
        // W must cover round_up(units, 128) rows (the host pads to 256)
        const int64_t wg128 = (int64_t)ktf_cdiv(d->units, 128) * ktf_cdiv(Tout, 128) * B;
        const bool lat = !(d->flags & KTF_TDNN_REF_TILES);   // flag: the register-staged 32x32x2 tile kernels (bitwise reference of the DMA-staged ones)
        if (lat && B * Tout <= 8) {
            dim3 grid((unsigned)ktf_cdiv(d->units, RV_UNITS), (unsigned)Tout, (unsigned)B);
            KTF_NOTE_KERNEL("tdnn_f32_rowvec_kernel");
            hipLaunchKernelGGL(tdnn_f32_rowvec_kernel, grid, dim3(64), 0, st, p);
        } else if (lat && wg128 < 256) {
            const int64_t wg64 = (int64_t)ktf_cdiv(d->units, FS_BM) * ktf_cdiv(Tout, FS_BM) * B;
#define FS_LAUNCH(BK_, BN_, NB_)                                                                                       \
    do {                                                                                                               \
        const int lds = FS_NSTAGE * (FS_BM + BN_) * BK_ * 4;                                                           \
        dim3 grid_((unsigned)ktf_cdiv(d->units, BN_), (unsigned)ktf_cdiv(Tout, FS_BM), (unsigned)B);                   \
        KTF_NOTE_KERNEL("tdnn_f32s_kernel<" #BK_ ", " #BN_ ">");                                                        \
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
                if (best == 32) FS_LAUNCH(64, 32, 1);
                else if (best == 64) FS_LAUNCH(64, 64, 1);
                else FS_LAUNCH(64, 96, 3);
            } else {
                FS_LAUNCH(32, 64, 1);
            }
#undef FS_LAUNCH
        } else if (lat) {
            dim3 grid((unsigned)ktf_cdiv(d->units, FT_BM), (unsigned)ktf_cdiv(Tout, FT_BM), (unsigned)B);
            KTF_NOTE_KERNEL("tdnn_f32t_kernel");
            KTF_LDS_ONCE(FT_LDS_BYTES, tdnn_f32t_kernel<32>);
            hipLaunchKernelGGL(tdnn_f32t_kernel<32>, grid, dim3(512), FT_LDS_BYTES, st, p);
        } else if (wg128 >= 256) {
            dim3 grid((unsigned)ktf_cdiv(d->units, 128), (unsigned)ktf_cdiv(Tout, 128), (unsigned)B);
            KTF_NOTE_KERNEL("tdnn_f32_kernel<2, 16>");
            hipLaunchKernelGGL((tdnn_f32_kernel<2, 16>), grid, dim3(256), 0, st, p);
        } else {
            dim3 grid((unsigned)ktf_cdiv(d->units, 64), (unsigned)ktf_cdiv(Tout, 64), (unsigned)B);
            KTF_NOTE_KERNEL("tdnn_f32_kernel<1, 32>");
            hipLaunchKernelGGL((tdnn_f32_kernel<1, 32>), grid, dim3(256), 0, st, p);
        }
    KTF_CHECK_LAUNCH("ktf_tdnn");
    return KTF_OK;
}
