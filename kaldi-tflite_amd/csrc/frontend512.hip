// Register-resident fast path of the fused front-end for nfft = 512 (the 25 ms / 16 kHz configuration of every golden
// and of the 0008_sitw_v2_1a model). Same arithmetic as frontend.hip, restructured so that one frame costs a fraction
// of the vector instructions and no LDS traffic for the FFT:
//   * the 512-point real FFT is a 256-point complex radix-4 FFT held ENTIRELY in registers (4 complex values per lane);
//     the three inter-stage data movements are 4x4 register/lane transposes done with cross-lane VALU moves
//     (v_permlane32_swap / v_permlane16_swap for lane bits 5 and 4, DPP row_ror:8 / row_shl|shr:4 with bank masks for
//     bits 3 and 2, DPP quad_perm for bits 1 and 0) — no LDS, no address arithmetic; twiddles are per-lane constants;
//   * one LDS round trip puts the spectrum in natural order for the real-FFT split (X[k], X[256-k]);
//   * the sparse mel bank is spread over all 64 lanes as (filter, <=16-bin slice) work items and a 2-step segmented
//     reduction; each lane computes one cepstral coefficient (its DCT column);
//   * the per-lane mel weights (16) and DCT column (32) are kept in LDS as lane-linear 16-byte records, not in registers:
//     12 more conflict-free ds_read_b128 per frame buy 48 VGPRs, i.e. 5 waves per SIMD instead of 3 — the kernel is bound
//     by the latency of its dependent DPP / permlane / LDS chains, not by VALU throughput, so resident waves are what it needs.
// Used by ktf_frontend_f32 when the caller provides the KtfFrontendTables.fast_* tables; any other configuration runs
// the generic kernel of frontend.hip.
#include "common.h"
#include <type_traits>

// Measured on MI355X (1024 x 998 frames, tools/ab_f5.sh; VGPRs / waves per SIMD of the hot <no dither, fp32 input, 400> instance):
//   constants in registers, 4 waves per workgroup                 136 VGPR, 3 waves/SIMD   0.945 ms   (round 1)
//   mel weights + DCT column in LDS                                122 VGPR, 4 waves/SIMD   0.876 ms
//   + twiddles in LDS, 4 waves per workgroup                        82 VGPR, 5 waves/SIMD   0.90  ms   (LDS-limited: 5 workgroups)
//   + 8 waves per workgroup (tables shared by twice the waves)      80 VGPR, 6 waves/SIMD   0.854 ms   <- this build
// 0.854 ms = 2040 cycles per frame and SIMD for ~490 vector instructions: 4.2 cycles per instruction, i.e. the VALU issue
// rate of this DPP / permlane / transcendental mix; more resident waves no longer help. Every later step removed instructions:
//   rounds 2-4 (register pairs, packed complex products, DPP sums and swaps, compile-time configuration)               0.659 ms
//   round 5: one-instruction row-broadcast adds                                                                          0.597 ms
//            DCT on two half-waves (16 + 2 instead of 32), mel segment sums as DPP multiply-adds (3 instead of 10),
//            no arithmetic on the register past the frame's last sample (-7), logf without its denormal branch (-6),
//            the spectrum's quarter in the mel weights (-4): ~330 issue slots per frame                                  0.549 ms
//   round 6: the LDS side (PMC: the LDS array was busy 81 % of the launch, 38 % of those cycles bank conflicts, SQ_WAIT_INST_LDS 22 % of the
//            wave cycles): the spectrum under an XOR swizzle of its index (the 8-byte writes were 4-way conflicts; per-lane constants only), the
//            mel items' bins as aligned 16-byte pieces with shifted weights (five reads issued together instead of <= 16 four-byte reads at
//            per-lane starts, each behind a wait of its own), 3 / 4 / 5 pieces by table for the shipped configuration: conflicts - 65 %,
//            LDS-active cycles - 25 %, SQ_WAIT_INST_LDS - 73 %, every MFCC bit the same                                   0.545 -> 0.525 ms (same box)
#ifndef F5_WAVES
#define F5_WAVES 8
#endif
#define F5_WAVE_FLOATS (2 * 256 + 64)   // per-wave LDS: Z[256] float2 (reused for P[256] float once the spectrum is split) | feat[64]
#define F5_THREADS (F5_WAVES * KTF_WAVE)
#define F5_MAXW 16          // bins per mel work item (upper bound; the table says how many are used)
#define F5_MELQ 5           // 16-byte pieces of the power spectrum a work item reads: its <= 16 bins start at any bin, the pieces at a multiple of 4
#define F5_MAXMEL 32        // DCT rows per lane
#ifndef F5_MINWAVES
#define F5_MINWAVES 6       // waves per SIMD the register allocation of the hot instance is held to (__launch_bounds__); the
                            // dither / int16 / mirror-padding instances keep 4 (they would spill in the frame loop)
#endif
#ifndef F5_TW_LDS
#define F5_TW_LDS 1         // 1: the per-lane FFT twiddles (26 floats) also live in LDS records instead of registers
#endif
#define F5_TWREC 28         // floats per lane in the twiddle record (7 x 16 B): tw1[3] tw2[3] tw3[3] rw[4] (+2 pad)

// ---- cross-lane primitives (all VALU: no LDS pipe)
#define DPP_QUAD_XOR1 0xB1      // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E      // quad_perm:[2,3,0,1]
#define DPP_ROW_SHL4 0x104
#define DPP_ROW_SHR4 0x114
#define DPP_ROW_ROR4 0x124
#define DPP_ROW_ROR8 0x128
#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138
#define DPP_WAVE_ROR1 0x13C

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// lanes selected by BANK_MASK (banks = groups of 4 lanes inside each row of 16) take `src` moved by CTRL, the rest keep `old`
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_merge(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, BANK_MASK, false));
}

#define DPP_ROW_BCAST15 0x142   // lane 15 of each row -> every lane of the next row (gfx9 wave64 DPP)
#define DPP_ROW_BCAST31 0x143   // lane 31 -> every lane of rows 2 and 3
// wave-wide sum, returned wave-uniform (an SGPR operand for the consumers): four row-local DPP steps, then the classic
// row_bcast:15 / row_bcast:31 combine (rows 1,3 += row 0,2; rows 2,3 += lane 31) and one v_readlane of lane 63 -- no
// ds_bpermute and none of its address arithmetic
__device__ __forceinline__ float wave_sum_f(float v) {
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_ROR4>(v);
    v += dpp_mov<DPP_ROW_ROR8>(v);           // every lane: sum of its row of 16
    // rows 1, 3 += lane 15 of rows 0, 2; rows 2, 3 += lane 31: one v_add_f32_dpp each, written under the row mask (the rows outside it keep
    // their value: from the builtin the compiler builds v_mov 0 / v_mov_dpp / v_add, three instructions per step, adding a zero to the
    // masked rows -- the same bits). The s_nop covers the DPP read-after-write hazard the compiler cannot see inside an asm statement.
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// logf for an argument that is a normal number or +inf (here: a non-negative sum + eps with eps >= FLT_MIN, which the kernel tests:
// KtfFrontendCfg.eps is FLT_EPSILON in every shipped configuration; any other eps takes logf itself): the bits of libm's logf -- v_log_f32 (log2, 1 ulp) times
// ln 2 as a compensated product -- without the rescaling of denormal arguments that logf carries (6 of its 18 instructions; a wave
// spends them once per frame on the log-energy).
__device__ __forceinline__ float log_normal(float x) {
    const float l = __builtin_amdgcn_logf(x);
    const float c = __uint_as_float(0x3f317217u), c_lo = __uint_as_float(0x3377d1cfu);
    const float p = c * l;
    float r = fmaf(l, c, -p);
    r = fmaf(c_lo, l, r);
    r = fmaf(c, l, r);
    return (fabsf(l) < __builtin_inff()) ? r : l;
}

// Complex product on the packed fp32 pipe: two instructions, the operand swizzles and the sign in the op_sel / neg modifiers (from C
// the compiler builds the same product from v_pk_mul + scalar v_mul / v_sub / v_add and two to four register moves).
//   t = (a.x b.x, a.y b.x);   r = (-a.y b.y + t.x, a.x b.y + t.y)
// Complex values live in ONE 64-bit register pair from the loads to the LDS write (an ext-vector, not HIP's float2 struct: as two
// scalars the compiler re-formed the pairs the packed instructions need with ~50 register moves per frame -- a seventh of the
// kernel's instructions; the cross-lane steps below read and write the halves of the pairs in place).
typedef float f5v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f5v2 cmulf(f5v2 av, f5v2 bv) {
    f5v2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[0,0,0]" : "=&v"(r) : "v"(av), "v"(bv));
    return r;
}
__device__ __forceinline__ f5v2 cadd(f5v2 a, f5v2 b) { return a + b; }
__device__ __forceinline__ f5v2 csub(f5v2 a, f5v2 b) { return a - b; }

// radix-4 DIF butterfly with W4 = -i:  y_r = sum_k z_k (-i)^{kr}
__device__ __forceinline__ void bfly4(f5v2 (&z)[4]) {
    const f5v2 apc = z[0] + z[2], amc = z[0] - z[2], bpd = z[1] + z[3], bmd = z[1] - z[3];
    // i (b - d) = (-bmd.y, bmd.x) never exists as a value: the two sums that use it take bmd with its halves exchanged (op_sel) and
    // one of them negated, one packed instruction each (built from C the compiler spends an xor and a move per butterfly on it)
    f5v2 y1, y3;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(y1) : "v"(amc), "v"(bmd));   // (amc.x + bmd.y, amc.y - bmd.x)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(y3) : "v"(amc), "v"(bmd));   // (amc.x - bmd.y, amc.y + bmd.x)
    z[0] = apc + bpd;
    z[1] = y1;
    z[2] = apc - bpd;
    z[3] = y3;
}

// One step of a 4x4 register/lane transpose: lanes whose bit BIT is 0 hand register `b` to their partner (lane ^ 2^BIT)
// and receive the partner's register `a` into `b`; lanes whose bit is 1 hand `a` and receive into `a`.
template <int BIT>
__device__ __forceinline__ void swap_step(float& a, float& b, int lane) {
    if constexpr (BIT == 5) {
        // v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]);
        b = __uint_as_float(r[1]);
    } else if constexpr (BIT == 4) {
        // v_permlane16_swap: odd rows (of 16 lanes) of the first operand <-> even rows of the second
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]);
        b = __uint_as_float(r[1]);
    } else if constexpr (BIT == 3) {
        const float na = dpp_merge<DPP_ROW_ROR8, 0xC>(a, b);     // lanes 8-15 of a row: a <- partner's b
        const float nb = dpp_merge<DPP_ROW_ROR8, 0x3>(b, a);     // lanes 0-7:           b <- partner's a
        a = na; b = nb;
    } else if constexpr (BIT == 2) {
        const float na = dpp_merge<DPP_ROW_SHR4, 0xA>(a, b);     // lanes 4-7,12-15: a <- b of lane-4
        const float nb = dpp_merge<DPP_ROW_SHL4, 0x5>(b, a);     // lanes 0-3,8-11:  b <- a of lane+4
        a = na; b = nb;
    } else {
        static_assert(BIT >= 2, "lane bits 0 and 1 go through swap4_quad");
    }
}

// swap_step<BIT> for BIT = 0, 1 on four register pairs at once. Inside a quad a DPP move cannot address single lanes (bank masks
// select groups of four), so the step is a select -- but v_cndmask_b32 takes its first operand THROUGH the DPP crossbar:
//   a' = (lane bit set) ? quad_perm(b) : a        b' = (lane bit set) ? b : quad_perm(a)
// is two instructions per pair (select-move-select-select was four). VOP2 + DPP reads its condition from VCC only, hence the
// assembly block: VCC = lanes with the bit clear, four selects, VCC = lanes with the bit set, four selects. The s_nop covers the
// DPP hazard (a VGPR written by the previous vector instruction is not readable through DPP for two cycles), which the compiler
// cannot see inside an asm statement.
template <int BIT>
__device__ __forceinline__ void swap4_quad(float& a0, float& b0, float& a1, float& b1, float& a2, float& b2, float& a3, float& b3) {
    static_assert(BIT == 0 || BIT == 1, "quad-local lane bits");
    float na0, na1, na2, na3;                     // (the b' are written in place: the second group reads a and b before it writes b)
#define F5_SWAP4(LO_, HI_, QP_)                                                                                        \
    asm("s_mov_b32 vcc_lo, " LO_ "\n\ts_mov_b32 vcc_hi, " LO_ "\n\ts_nop 1\n\t"                                        \
        "v_cndmask_b32_dpp %0, %4, %8, vcc " QP_ " row_mask:0xf bank_mask:0xf\n\t"                                     \
        "v_cndmask_b32_dpp %1, %5, %9, vcc " QP_ " row_mask:0xf bank_mask:0xf\n\t"                                     \
        "v_cndmask_b32_dpp %2, %6, %10, vcc " QP_ " row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_cndmask_b32_dpp %3, %7, %11, vcc " QP_ " row_mask:0xf bank_mask:0xf\n\t"                                    \
        "s_mov_b32 vcc_lo, " HI_ "\n\ts_mov_b32 vcc_hi, " HI_ "\n\t"                                                   \
        "v_cndmask_b32_dpp %4, %8, %4, vcc " QP_ " row_mask:0xf bank_mask:0xf\n\t"                                     \
        "v_cndmask_b32_dpp %5, %9, %5, vcc " QP_ " row_mask:0xf bank_mask:0xf\n\t"                                     \
        "v_cndmask_b32_dpp %6, %10, %6, vcc " QP_ " row_mask:0xf bank_mask:0xf\n\t"                                    \
        "v_cndmask_b32_dpp %7, %11, %7, vcc " QP_ " row_mask:0xf bank_mask:0xf"                                         \
        : "=&v"(na0), "=&v"(na1), "=&v"(na2), "=&v"(na3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)                       \
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3)                                                                           \
        : "vcc")
    if constexpr (BIT == 0) F5_SWAP4("0x55555555", "0xaaaaaaaa", "quad_perm:[1,0,3,2]");
    else F5_SWAP4("0x33333333", "0xcccccccc", "quad_perm:[2,3,0,1]");
#undef F5_SWAP4
    a0 = na0; a1 = na1; a2 = na2; a3 = na3;
}

// 4x4 transpose between the 4 registers of a lane and the 4 lanes that differ only in lane bits (BH, BL):
// afterwards the lane at group position p = 2*bit(BH) + bit(BL) holds in register q what position q held in register p.
template <int BH, int BL>
__device__ __forceinline__ void transpose4(f5v2 (&r)[4], int lane) {
#define F5_SS(B_, p_, q_, c_) { float a_ = r[p_].c_, b_ = r[q_].c_; swap_step<B_>(a_, b_, lane); r[p_].c_ = a_; r[q_].c_ = b_; }
#define F5_SQ(B_, p0, q0, p1, q1) { float a0 = r[p0].x, b0 = r[q0].x, a1 = r[p0].y, b1 = r[q0].y, a2 = r[p1].x, b2 = r[q1].x, a3 = r[p1].y, b3 = r[q1].y; \
        swap4_quad<B_>(a0, b0, a1, b1, a2, b2, a3, b3); r[p0].x = a0; r[q0].x = b0; r[p0].y = a1; r[q0].y = b1; r[p1].x = a2; r[q1].x = b2; r[p1].y = a3; r[q1].y = b3; }
    if constexpr (BH <= 1) {
        F5_SQ(BH, 0, 2, 1, 3)
    } else {
        F5_SS(BH, 0, 2, x) F5_SS(BH, 0, 2, y) F5_SS(BH, 1, 3, x) F5_SS(BH, 1, 3, y)
    }
    if constexpr (BL <= 1) {
        F5_SQ(BL, 0, 1, 2, 3)
    } else {
        F5_SS(BL, 0, 1, x) F5_SS(BL, 0, 1, y) F5_SS(BL, 2, 3, x) F5_SS(BL, 2, 3, y)
    }
#undef F5_SS
#undef F5_SQ
}

// Dither (windowing.py:182-183: x += N(0,1) * dither): counter-based Philox4x32-10 keyed by `seed`, counter (frame, lane). ONE call
// per lane and frame: its four 32-bit words are eight 16-bit uniforms = four Box-Muller pairs = the EIGHT Gaussians of the lane's eight
// samples (radius from v_log_f32 / v_sqrt_f32, angle straight into v_sin_f32 / v_cos_f32, whose argument is in revolutions). 16-bit
// uniforms bound the noise at 4.85 sigma in steps of <= 3e-5 sigma -- dither, not a Monte-Carlo source (Kaldi's own RandGauss draws from
// rand()). The 32-bit multiplies of Philox run at a quarter of the vector rate, so halving the calls is what counts: two calls and the
// exact sinpif / cospif took 1.2 ms per 1024 x 998 frames, this form takes half (the generic kernel in frontend.hip still spends one
// call and one log / sqrt / cos per sample).
__device__ __forceinline__ void gauss_noise8(uint64_t seed, uint64_t row, uint32_t i, float (&g)[8]) {
    uint32_t c0 = (uint32_t)row, c1 = (uint32_t)(row >> 32), c2 = i, c3 = 0x9E3779B9u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float u1 = fmaf((float)(c[k] & 0xffffu), 1.0f / 65536.0f, 0.5f / 65536.0f);
        const float u2 = fmaf((float)(c[k] >> 16), 1.0f / 65536.0f, 0.5f / 65536.0f);
        const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // sqrt(-2 ln u1), v_log_f32 = log2
        g[2 * k] = ra * __builtin_amdgcn_cosf(u2);                                                        // cos(2 pi u2)
        g[2 * k + 1] = ra * __builtin_amdgcn_sinf(u2);
    }
}

#define F5_WAVE_SYNC()                           \
    do {                                         \
        asm volatile("" ::: "memory");           \
        __builtin_amdgcn_wave_barrier();         \
        asm volatile("" ::: "memory");           \
    } while (0)

// KIND 1: fp32 input without mirror padding (the hot path keeps its scalar-register budget); KIND 2: the same for int16
// PCM (what a deployment uploads: half the PCIe bytes) -- eight 16-bit loads per lane instead of eight dword loads, nothing
// else differs, so it keeps the hot instance's registers and occupancy; KIND 0 (general) adds int16 samples behind a run-time switch
// and KtfFrontendCfg.pad_mode.
// MFIX: frame size known at compile time (400 = 25 ms at 16 kHz, the shipped configuration) or 0 = cfg.frame_size: with
// a constant M the `sample index < M` predicates of the loads, DC removal and pre-emphasis fold away.
// STD: the configuration of the shipped models (MFCC's defaults, data/kaldi_models/configs/*.yml) known at compile time -- waveform in,
// MFCC out, DC removal, raw log-energy into C0, pre-emphasis, power spectrum, log mel: the run-time tests of those switches, and the
// register copies the compiler keeps for values a skipped branch would have left unchanged, fold away (~25 instructions per frame).
template <bool DITHER, int KIND, int MFIX, bool STD = false, int NQ = F5_MELQ>
__global__ __launch_bounds__(F5_THREADS, (KIND != 0 && !DITHER) ? F5_MINWAVES : 4) void frontend512_kernel(const void* __restrict__ in_v, int64_t B, int64_t n,
                                                                 int in_kind, KtfFrontendCfg cfg, KtfFrontendTables tab,
                                                                 int out_stage, float* __restrict__ out,
                                                                 uint64_t seed, int64_t T) {
    constexpr int NF = 512, N2 = 256, NV = 8;
    // registers that hold samples of the frame: with the frame size known (400: v[7] = samples 448.. holds none) the last register is
    // zero by construction and its share of the sums, the pre-emphasis and the window multiply is never issued (its pre-emphasis
    // neighbour, lane 63 of v[6], is beyond the frame too: zero before and after)
    constexpr int NVA = MFIX ? (MFIX + KTF_WAVE - 1) / KTF_WAVE : NV;
    constexpr bool PLAIN = KIND != 0;
    extern __shared__ __attribute__((aligned(16))) float lds5[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = MFIX ? MFIX : cfg.frame_size;
    const int nm = cfg.num_mels, nc = cfg.num_ceps;
    if (STD) { in_kind = (KIND == 2) ? KTF_IN_WAV_I16 : KTF_IN_WAV; out_stage = KTF_OUT_MFCC; }
    const bool c_dc = STD ? true : cfg.remove_dc != 0;
    const bool c_energy = STD ? true : cfg.use_energy != 0;
    const bool c_raw_e = STD ? true : (cfg.use_energy && cfg.raw_energy);
    const bool c_post_e = STD ? false : (cfg.use_energy && !cfg.raw_energy);
    const bool c_pre = STD ? true : cfg.preemph > 0.0f;
    const bool c_pow = STD ? true : cfg.use_power != 0;
    const bool c_log = STD ? true : cfg.use_log != 0;

    // LDS: window[512] | dct4[8][64] f32x4 | melw4[4][64] f32x4 | (twiddle records) | per wave: Z[256] float2 = P[256] float, feat[64] float
    float* win = lds5;
    float* dct4 = lds5 + NF;                                   // record (m4, lane) = dct[4 m4 .. 4 m4 + 3][lane]
    float* melw4 = dct4 + F5_MAXMEL * KTF_WAVE;                // record (j4, lane) = this lane's mel weights 4 j4 .. 4 j4 + 3
    float* twl = melw4 + F5_MELQ * 4 * KTF_WAVE;                // F5_TW_LDS: record (q, lane), q < 7
    float* wbase = twl + (F5_TW_LDS ? F5_TWREC * KTF_WAVE : 0) + wave * F5_WAVE_FLOATS;
    float2* Zb = reinterpret_cast<float2*>(wbase);
    float* Pb = wbase;               // the power spectrum overwrites Z: a wave's LDS operations execute in program order, and
                                     // every Z read of the split precedes the first P write
    float* feat = wbase + 2 * N2;
    if (in_kind != KTF_IN_WINDOWED)
        for (int i = tid; i < NF; i += F5_THREADS) win[i] = (i < M) ? tab.window[i] : 0.0f;
    feat[lane] = 0.0f;

    // ---- per-lane constants (registers for the whole kernel)
#if F5_TW_LDS
    for (int i = tid; i < F5_TWREC * KTF_WAVE; i += F5_THREADS) {
        const int e = i & 3, l = (i >> 2) & 63, f = 4 * (i >> 8) + e;       // float f of lane l's record
        float val = 0.0f;
        if (f < 18) val = tab.fast_tw[l * 18 + f];
        else if (f < 26) val = tab.rtwiddle[2 * (l + 64 * ((f - 18) >> 1)) + ((f - 18) & 1)];
        twl[i] = val;
    }
#define F5_TWQ(q) (*reinterpret_cast<const f32x4*>(twl + ((q) * KTF_WAVE + lane) * 4))
#else
    float2 tw1[3], tw2[3], tw3[3], rw[4];
    {
        const float* t = tab.fast_tw + lane * 18;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            tw1[r] = make_float2(t[2 * r], t[2 * r + 1]);
            tw2[r] = make_float2(t[6 + 2 * r], t[6 + 2 * r + 1]);
            tw3[r] = make_float2(t[12 + 2 * r], t[12 + 2 * r + 1]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            rw[j] = make_float2(tab.rtwiddle[2 * k], tab.rtwiddle[2 * k + 1]);
        }
    }
#endif
    const int mel_start = tab.fast_mel_meta[lane * 4 + 0];
    const int mel_filter = tab.fast_mel_meta[lane * 4 + 2];
    const int mel_flags = tab.fast_mel_meta[lane * 4 + 3];      // bit0: lane+1 same filter, bit1: lane+2 same, bit2: first
    const float mel_f1 = (mel_flags & 1) ? 1.0f : 0.0f, mel_f2 = (mel_flags & 2) ? 1.0f : 0.0f;
    // |X|^2 = |2 E + rw 2 O|^2 / 4 (below): the quarter lives in the LDS copy of the mel weights, not in a multiply per bin (a power of
    // two: every product and every partial sum is the same number either way; the magnitude spectrum carries the half of it)
    const float mel_scale = c_pow ? 0.25f : 0.5f;
    // A work item's bins are read as ALIGNED 16-byte pieces of the power spectrum (five cover any 16 bins) and its weights are stored shifted
    // by (first bin & 3), zeros in front and behind: the products that matter are the same, in the same order (a zero weight leaves the sum's
    // bits alone), and the 16 four-byte reads at per-lane starts -- each behind a wait of its own, 2-way bank conflicts on average: the other
    // half of the kernel's LDS conflict cycles -- become five reads issued together.
    for (int i = tid; i < F5_MELQ * 4 * KTF_WAVE; i += F5_THREADS) {      // weights are zero beyond an item's length
        const int e = i & 3, l = (i >> 2) & 63, q = i >> 8;
        const int idx = 4 * q + e - (tab.fast_mel_meta[l * 4 + 0] & 3);
        melw4[i] = (idx >= 0 && idx < F5_MAXW) ? mel_scale * tab.fast_mel_w[l * F5_MAXW + idx] : 0.0f;
    }
    float lift = 1.0f;
    if (out_stage == KTF_OUT_MFCC) {
        for (int i = tid; i < F5_MAXMEL * KTF_WAVE; i += F5_THREADS) {
            const int e = i & 3, l = (i >> 2) & 63, m = 4 * (i >> 8) + e;
            dct4[i] = (m < nm && l < nc) ? tab.dct[m * nc + l] : 0.0f;
        }
        if (cfg.use_lifter && tab.lifter && lane < nc) lift = tab.lifter[lane];
    }
    const bool eps_normal = cfg.eps >= 1.17549435e-38f;        // wave-uniform: the log-energy's argument is then a normal number
    const bool dct_halves = nc <= 32;                           // wave-uniform
    const int dct_row0 = 16 * (lane >> 5);
    const float* dct_col = dct4 + ((4 * (lane >> 5)) * KTF_WAVE + (lane & 31)) * 4;
    // output index of this lane's FFT results: X[mo + 64*r4]
    const int mo = (2 * (lane & 1) + ((lane >> 5) & 1)) + 4 * ((lane >> 3) & 3) + 16 * ((lane >> 1) & 3);
    // Z lives in LDS under an XOR swizzle of its index, bits 4 and 5 folded into bits 0 and 3 (F5_ZSW): the 16 lanes that one LDS cycle of the
    // 8-byte writes serves differ in mo's bits 1, 2, 4, 5 -- dword bits 2, 3, 5, 6, of which the bank (dword index mod 32) sees two: a 4-way
    // conflict on every write of the spectrum (PMC, round 5: 38 % of the kernel's LDS cycles were conflict cycles, the LDS array busy 81 % of
    // the launch -- as busy as the vector pipe). The swizzle term depends on index bits below 6 only: for the writer (mo + 64 r) and for both
    // readers (k = lane + 64 j and its mirror 256 - k, whose low six bits are the lane's alone) it is a per-lane constant -- no instruction in
    // the frame loop changes, the same numbers travel, and each of the three access patterns touches 16 (32) distinct bank pairs.
#define F5_ZSW(k_) ((k_) ^ (((k_) >> 4) & 1) ^ ((((k_) >> 5) & 1) << 3))
    const int mo_s = F5_ZSW(mo), zk_s = F5_ZSW(lane);
    const float invM = 1.0f / (float)M;
    __syncthreads();

    // grid = (frame groups per utterance, B): no division in the frame loop, 32-bit offsets from per-utterance bases
    const int b = blockIdx.y;
    const int Ti = (int)T;
    const int i16 = (KIND == 2) || (!PLAIN && in_kind == KTF_IN_WAV_I16);
    if (in_kind == KTF_IN_WAV_I16) in_kind = KTF_IN_WAV;
    const int64_t rstride = cfg.row_stride > 0 ? (int64_t)cfg.row_stride : n;
    const float* in_b = reinterpret_cast<const float*>(in_v) + (int64_t)b * ((in_kind == KTF_IN_WAV) ? rstride : T * (int64_t)M);
    const short* in_b16 = reinterpret_cast<const short*>(in_v) + (int64_t)b * rstride;
    const int src_step = (in_kind == KTF_IN_WAV) ? cfg.frame_shift : M;
    const int pad_left = (!PLAIN && in_kind == KTF_IN_WAV && cfg.pad_mode) ? (M - cfg.frame_shift) / 2 : 0;
    const int ni = (int)n;
    const int64_t row_base = (int64_t)b * T;
    const int t_step = gridDim.x * F5_WAVES;
    for (int t0 = blockIdx.x * F5_WAVES; t0 < Ti; t0 += t_step) {
        const int t = t0 + wave;
        if (t >= Ti) continue;       // wave-uniform; nothing below synchronises across waves (wave-level barriers only)
        constexpr bool valid = true;
        const int64_t row = row_base + t;
        float logE = 0.0f;
        const int g0 = t * src_step - pad_left;          // first sample of the frame (wav kinds)
        f5v2 z[4];
        {
        float v[NV];
        if (!PLAIN && valid && in_kind == KTF_IN_WAV && (g0 < 0 || g0 + M > ni)) {
            // edge frame of KtfFrontendCfg.pad_mode 1: mirrored samples
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + KTF_WAVE * j;
                v[j] = (i < M) ? ktf_wav_sample(i16 ? (const void*)in_b16 : (const void*)in_b, i16, ni, g0 + i) : 0.0f;
            }
        } else if (!PLAIN && valid && i16) {
            const short* src = in_b16 + g0;
            if (((reinterpret_cast<uintptr_t>(src) | (unsigned)M) & 3u) == 0 || ((reinterpret_cast<uintptr_t>(src) & 3u) == 0 && (M & 1) == 0)) {
                // dword loads: lanes 2k, 2k+1 fetch the same word. All loads are issued before the first conversion (a
                // conversion inside the predicated load would put an s_waitcnt vmcnt(0) behind every load).
                const unsigned* src32 = reinterpret_cast<const unsigned*>(src);
                const int sh = 16 - (lane & 1) * 16;
                unsigned wd[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    wd[j] = (i < M) ? src32[i >> 1] : 0u;
                }
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = (float)((int)(wd[j] << sh) >> 16);
            } else {
                int raw[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    raw[j] = (i < M) ? (int)src[i] : 0;
                }
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = (float)raw[j];
            }
        } else if (KIND == 2 && valid) {
            const short* src = in_b16 + g0;
            int raw[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + KTF_WAVE * j;
                raw[j] = (i < M) ? (int)src[i] : 0;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = (float)raw[j];
        } else if (valid) {
            const float* src = in_b + g0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + KTF_WAVE * j;
                v[j] = (i < M) ? src[i] : 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = 0.0f;
        }

        // ---- Windowing.call (windowing.py:180-209)
        if (in_kind != KTF_IN_WINDOWED) {
            if (DITHER) {
                float g[NV];                                          // the eight samples of this lane from one Philox call
                gauss_noise8(seed, (uint64_t)row, (uint32_t)lane, g);
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (lane + KTF_WAVE * j < M) v[j] += g[j] * cfg.dither;
            }
            if (c_dc) {
                float s = v[0];
#pragma unroll
                for (int j = 1; j < NVA; ++j) s += v[j];
                const float mean = wave_sum_f(s) * invM;
#pragma unroll
                for (int j = 0; j < NVA; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    if (i < M) v[j] -= mean;
                }
            }
            if (c_raw_e) {
                float s = v[0] * v[0];
#pragma unroll
                for (int j = 1; j < NVA; ++j) s = fmaf(v[j], v[j], s);
                {
                    const float en = fmaxf(wave_sum_f(s), 0.0f) + cfg.eps;
                    logE = fmaxf(eps_normal ? log_normal(en) : logf(en), cfg.energy_floor);
                }
            }
            if (c_pre) {
                // y[i] = x[i] - c x[i-1] (x[-1] := x[0]). Sample i - 1 lives in lane l - 1 of the same register, for lane 0 in lane 63
                // of the previous one: a whole-wave rotate of v[j-1] puts that value into lane 0, where the whole-wave shift of v[j]
                // (which has no source for lane 0) leaves it -- two DPP moves per register, no readlane / select
                float y[NV];
#pragma unroll
                for (int j = NVA; j < NV; ++j) y[j] = 0.0f;
#pragma unroll
                for (int j = 0; j < NVA; ++j) {
                    const int wrap = (j > 0) ? __builtin_amdgcn_update_dpp(0, __float_as_int(v[j > 0 ? j - 1 : 0]), DPP_WAVE_ROR1, 0xF, 0xF, true)
                                             : __float_as_int(v[0]);
                    const float prev = __int_as_float(__builtin_amdgcn_update_dpp(wrap, __float_as_int(v[j]), DPP_WAVE_SHR1, 0xF, 0xF, false));
                    y[j] = v[j] - cfg.preemph * prev;
                }
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = y[j];
            }
#pragma unroll
            for (int j = 0; j < NVA; ++j) v[j] *= win[lane + KTF_WAVE * j];       // window is zero beyond M
            if (c_post_e) {
                float s = v[0] * v[0];
#pragma unroll
                for (int j = 1; j < NVA; ++j) s = fmaf(v[j], v[j], s);
                {
                    const float en = fmaxf(wave_sum_f(s), 0.0f) + cfg.eps;
                    logE = fmaxf(eps_normal ? log_normal(en) : logf(en), cfg.energy_floor);
                }
            }
        }

        swap4_quad<0>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);    // even lanes: (own, partner's) v[2k]; odd: (partner's, own) v[2k+1]
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = f5v2{v[2 * k], v[2 * k + 1]};
        }

        // ---- 256-point complex FFT of z[n] = x[2n] + i x[2n+1], in registers.
        // lane l holds samples l + 64 j; after one lane^1 exchange it holds z[n0 + 64 k], n0 = (l>>1) + 32 (l&1)
#if F5_TW_LDS
        // record floats: tw1 = 0..5, tw2 = 6..11, tw3 = 12..17, rw = 18..25
        const f32x4 q0 = F5_TWQ(0), q1 = F5_TWQ(1), q2 = F5_TWQ(2), q3 = F5_TWQ(3), q4 = F5_TWQ(4);
        const f5v2 tw1[3] = {f5v2{q0.x, q0.y}, f5v2{q0.z, q0.w}, f5v2{q1.x, q1.y}};
        const f5v2 tw2[3] = {f5v2{q1.z, q1.w}, f5v2{q2.x, q2.y}, f5v2{q2.z, q2.w}};
        const f5v2 tw3[3] = {f5v2{q3.x, q3.y}, f5v2{q3.z, q3.w}, f5v2{q4.x, q4.y}};
#endif
        bfly4(z);                                   // over k (stride 64)
#pragma unroll
        for (int r = 1; r < 4; ++r) z[r] = cmulf(z[r], tw1[r - 1]);
        transpose4<0, 5>(z, lane);                  // group = lanes differing in bits {0, 5}  (n0 = n1 + 16 k2)
        bfly4(z);
#pragma unroll
        for (int r = 1; r < 4; ++r) z[r] = cmulf(z[r], tw2[r - 1]);
        transpose4<4, 3>(z, lane);                  // n1 = n2 + 4 k3
        bfly4(z);
#pragma unroll
        for (int r = 1; r < 4; ++r) z[r] = cmulf(z[r], tw3[r - 1]);
        transpose4<2, 1>(z, lane);                  // n2 = k4
        bfly4(z);
        // natural order through LDS (wave-private): Z[mo + 64 r4]
#pragma unroll
        for (int r = 0; r < 4; ++r) Zb[mo_s + 64 * r] = make_float2(z[r].x, z[r].y);
        F5_WAVE_SYNC();
        // ---- split the packed spectrum, |X[k]|(^2)  (filterbank.py:232-235; bin 256 carries no mel weight)
        float pw[4];
#if F5_TW_LDS
        const f32x4 q4b = F5_TWQ(4), q5 = F5_TWQ(5), q6 = F5_TWQ(6);
        const float2 rw[4] = {make_float2(q4b.z, q4b.w), make_float2(q5.x, q5.y), make_float2(q5.z, q5.w), make_float2(q6.x, q6.y)};
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            const float2 zk = Zb[zk_s + 64 * j], zm = Zb[F5_ZSW((N2 - k) & (N2 - 1))];
            // X[k] = E + rw O with E = (zk + conj zm) / 2, O = -i (zk - conj zm) / 2 = ((zk.y + zm.y), (zm.x - zk.x)) / 2. The halves are
            // factored out (powers of two: exact): |X|^2 = |2 E + rw 2 O|^2 / 4, each step one packed instruction
            const f5v2 kv = {zk.x, zk.y}, mv = {zm.x, zm.y}, rv = {rw[j].x, rw[j].y};
            f5v2 e2, o2, u, x2;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(e2) : "v"(kv), "v"(mv));                                   // (zk.x + zm.x, zk.y - zm.y)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(o2) : "v"(kv), "v"(mv));      // (zk.y + zm.y, zm.x - zk.x)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(u) : "v"(o2), "v"(rv), "v"(e2));               // e2 + o2 * rw.x
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[0,0,0]" : "=v"(x2) : "v"(o2), "v"(rv), "v"(u));   // (-o2.y rw.y, o2.x rw.y) + u
            // |X|^2 directly (what Kaldi's ComputePowerSpectrum does; the reference's abs-then-square differs by an ulp)
            pw[j] = fmaf(x2.x, x2.x, x2.y * x2.y);           // 4 |X|^2: the mel weights carry the quarter
        }
        if (!c_pow) {                        // wave-uniform branch: the magnitude spectrum pays for its sqrt only when asked
#pragma unroll
            for (int j = 0; j < 4; ++j) pw[j] = sqrtf(pw[j]);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) Pb[lane + 64 * j] = pw[j];
        F5_WAVE_SYNC();
        // ---- sparse mel bank: this lane's slice of one filter (weights are zero beyond the slice; the reads stay inside
        //      the wave's own P/feat area), then a segmented reduction over <= 4 adjacent lanes
        float acc = 0.0f;
        {
            const float* pq = Pb + (mel_start & ~3);           // (beyond bin 255 the pieces read this frame's own Z values: finite, times zero)
            // (no test per piece at run time: a wave-uniform test per piece compiled to a select, a compare and two branches each, and ONE
            // branch per frame around straight-line bodies of three / four / five pieces measured 2.7 % slower than five unconditional pieces,
            // 0.555 against 0.540 ms, with 18 more vector instructions per frame in the counters)
            // NQ: the shipped configuration's instances know how many pieces their table needs (the launcher picks 3, 4 or 5 from its longest item)
            f32x4 pv[NQ], wv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                pv[q] = *reinterpret_cast<const f32x4*>(pq + 4 * q);
                wv[q] = *reinterpret_cast<const f32x4*>(melw4 + (q * KTF_WAVE + lane) * 4);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                acc = fmaf(pv[q].x, wv[q].x, acc);
                acc = fmaf(pv[q].y, wv[q].y, acc);
                acc = fmaf(pv[q].z, wv[q].z, acc);
                acc = fmaf(pv[q].w, wv[q].w, acc);
            }
        }
        {   // segmented reduction over <= 4 adjacent lanes: lane i takes lane i + 1, then lane i + 2, by whole-wave DPP shifts (no
            // ds_bpermute and none of its address arithmetic). The per-lane "same filter" flags are 0.0 / 1.0 factors of a v_fmac_f32
            // whose first operand comes through the DPP crossbar: acc += flag * acc[lane + 1] is ONE instruction (select forms: zero,
            // move, select, add), the same bits for finite sums (x * 1 + acc rounds once, x * 0 + acc = acc). Lane 63 has no source
            // lane: its write is dropped, and the shifted copy gets a zero there (bound_ctrl) so that lane 62 multiplies a number.
            asm("s_nop 1\n\t"
                "v_fmac_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mel_f1));
            const float t2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), DPP_WAVE_SHL1, 0xF, 0xF, true));
            asm("s_nop 1\n\t"
                "v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t2), "v"(mel_f2));
        }
        // (v_log_f32 is log2 to 1 ulp; its argument here is >= eps, a normal number: none of logf's denormal handling is needed)
        if (c_log) acc = 0.6931471805599453f * __builtin_amdgcn_logf(fmaxf(acc, 0.0f) + cfg.eps);
        if (mel_flags & 4) {
            if (out_stage == KTF_OUT_FBANK) {
                if (valid) out[row * (int64_t)nm + mel_filter] = acc;
            } else {
                feat[mel_filter] = acc;
            }
        }
        if (out_stage == KTF_OUT_FBANK) continue;
        F5_WAVE_SYNC();
        // ---- DCT (this lane's column, coefficients in registers; log-mel vector broadcast from LDS) + lifter + C0
        float c = 0.0f;
        if (dct_halves) {
            // <= 32 coefficients: lanes 0-31 sum mel rows 0-15 of their column, lanes 32-63 rows 16-31 of the same columns, one
            // v_permlane32_swap brings the upper half's sums down: 16 + 2 vector instructions and 8 LDS reads instead of 32 and 16
            // (fp32 sum order: rows 0-15 and 16-31 separately, then added)
#pragma unroll
            for (int m4 = 0; m4 < F5_MAXMEL / 8; ++m4) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(feat + dct_row0 + 4 * m4);   // rows >= num_mels are zero
                const f32x4 d = *reinterpret_cast<const f32x4*>(dct_col + m4 * KTF_WAVE * 4);
                c = fmaf(f.x, d.x, c);
                c = fmaf(f.y, d.y, c);
                c = fmaf(f.z, d.z, c);
                c = fmaf(f.w, d.w, c);
            }
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(c), false, false);
            c += __uint_as_float(sw[1]);         // lanes 0-31: + lanes 32-63's sum (the upper lanes double theirs: never stored)
        } else {
#pragma unroll
            for (int m4 = 0; m4 < F5_MAXMEL / 4; ++m4) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(feat + 4 * m4);   // rows >= num_mels are zero
                const f32x4 d = *reinterpret_cast<const f32x4*>(dct4 + (m4 * KTF_WAVE + lane) * 4);
                c = fmaf(f.x, d.x, c);
                c = fmaf(f.y, d.y, c);
                c = fmaf(f.z, d.z, c);
                c = fmaf(f.w, d.w, c);
            }
        }
        c *= lift;
        if (lane == 0 && c_energy) c = logE;
        if (valid && lane < nc) out[row * (int64_t)nc + lane] = c;
        F5_WAVE_SYNC();
    }
}

// launcher used by ktf_frontend_f32 (frontend.hip) when the fast tables are present and the configuration qualifies
int ktf_frontend512_launch(const void* in, int64_t B, int64_t n, int32_t in_kind, const KtfFrontendCfg* cfg,
                           const KtfFrontendTables* tab, int32_t out_stage, float* out, uint64_t seed, int64_t T,
                           hipStream_t st) {
    // ~2048 workgroups in total: gx frame groups per utterance x B utterances
    KTF_REQUIRE(B < 65536 && T * (int64_t)512 < (1ll << 31) && n < (1ll << 31), "ktf_frontend_f32(fast512): B, T or n too large");
    int gx = (int)(2048 / (B > 0 ? B : 1));
    const int gmax = ktf_cdiv(T, F5_WAVES);
    if (gx < 1) gx = 1;
    if (gx > gmax) gx = gmax;
    const dim3 grid((unsigned)gx, (unsigned)B);
    const size_t lds = sizeof(float) * (512 + (F5_MAXMEL + F5_MELQ * 4 + (F5_TW_LDS ? F5_TWREC : 0)) * KTF_WAVE + F5_WAVES * F5_WAVE_FLOATS);
    const bool dither = cfg->dither != 0.0f && in_kind != KTF_IN_WINDOWED;
    const bool padded = (in_kind == KTF_IN_WAV || in_kind == KTF_IN_WAV_I16) && cfg->pad_mode;
    const int kind = padded ? 0 : (in_kind == KTF_IN_WAV_I16 ? 2 : 1);
#define F5_LAUNCH(DI, KI, MF)                                                                                          \
    hipLaunchKernelGGL((frontend512_kernel<DI, KI, MF>), grid, dim3(F5_THREADS), lds, st, in, B, n, in_kind, *cfg,    \
                       *tab, out_stage, out, seed, T)
    const bool std_cfg = !dither && !padded && cfg->frame_size == 400 && (in_kind == KTF_IN_WAV || in_kind == KTF_IN_WAV_I16) &&
                         out_stage == KTF_OUT_MFCC && cfg->remove_dc && cfg->use_energy && cfg->raw_energy && cfg->preemph > 0.0f &&
                         cfg->use_power && cfg->use_log;
    if (std_cfg) {
        // (tab->reserved = the longest mel work item; at any shift it spans (reserved + 3 + 3) / 4 aligned pieces: four for the 10-bin items
        // of the 30- and 40-mel banks at 16 kHz, five for the 13-bin items of 23 mels)
        const int nq = (tab->reserved + 6) / 4;
#define F5_STD(KI, NQ_) hipLaunchKernelGGL((frontend512_kernel<false, KI, 400, true, NQ_>), grid, dim3(F5_THREADS), lds, st, in, B, n, in_kind, *cfg, *tab, out_stage, out, seed, T)
        if (kind == 2) { if (nq <= 3) F5_STD(2, 3); else if (nq == 4) F5_STD(2, 4); else F5_STD(2, F5_MELQ); }
        else { if (nq <= 3) F5_STD(1, 3); else if (nq == 4) F5_STD(1, 4); else F5_STD(1, F5_MELQ); }
#undef F5_STD
        KTF_CHECK_LAUNCH("ktf_frontend_f32(fast512)");
        return KTF_OK;
    }
#define F5_KINDS(DI, MF)                                                                                               \
    do {                                                                                                               \
        if (kind == 1) F5_LAUNCH(DI, 1, MF);                                                                           \
        else if (kind == 2) F5_LAUNCH(DI, 2, MF);                                                                      \
        else F5_LAUNCH(DI, 0, MF);                                                                                     \
    } while (0)
    if (dither) F5_KINDS(true, 0);
    else if (cfg->frame_size == 400) F5_KINDS(false, 400);
    else F5_KINDS(false, 0);
#undef F5_KINDS
#undef F5_LAUNCH
    KTF_CHECK_LAUNCH("ktf_frontend_f32(fast512)");
    return KTF_OK;
}
