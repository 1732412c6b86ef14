// Shared by the KTF_GEMM_F16MX kernels (tdnn_mx.hip: 256 x 256 tile on eight waves, tdnn_mxl.hip: 192 x 256
// tile on eight matrix waves + four loader waves): operand types, the parameter block, the E8M0 scale rule.
#pragma once
#include "tdnn_common.h"
#include "flat_stats.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) _Float16 hfrag8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define MX_TILE 16384                       // one half operand image of a K-step: 256 rows x 64 B
#define MX_STAGE (2 * MX_TILE)              // A | W
#define MX_SA_OFF (2 * MX_STAGE)            // side A: xl4 [4][256][16] | x4 [4][256][16] | scales [4][256] u32
#define MX_SA_BYTES (2 * 16384 + 4096)
#define MX_SW_OFF (MX_SA_OFF + MX_SA_BYTES)  // side W: w4 [4][256][16] | wl6a [4][256][16] | wl6b [4][256][8] | scales [4][256] u32 | pad
#define MX_WQ_BLOCK 49152                   // bytes of one (N-tile, super-step) block of the MX weight planes (last 4 KiB unused)
#define MX_PRM_OFF (MX_SW_OFF + MX_WQ_BLOCK)        // bias | scale | shift of the tile's 256 columns (read by the epilogue)
#define MX_LDS_BYTES (MX_PRM_OFF + 3 * 256 * 4)     // 154,624 B
// tdnn_mx.hip: behind it the K-step table of the tile, two int32 arrays of (padded K-steps + 4) entries: (chunk * T + MX_KQ_BIAS) and the
// context offset of K-step k + 1 in entry k
#define MX_KQ_OFF MX_LDS_BYTES
#define MX_KQ_BIAS 128                              // records: the activation planes' buffer resources start this far in front of the plane
#define MX_KQ_MAX_STEPS 1144                        // padded K-steps a layer may have: 8 * (steps + 4) B of table must fit the CU's 160 KiB (K <= 36,608)
#define MX_LDS_TOTAL(nkp) (MX_LDS_BYTES + 8 * ((nkp) + 4))
static_assert(8 * 64 * 68 * 4 <= MX_PRM_OFF, "epilogue staging regions");
#define MX_EPW_PITCH 68                             // wave-private epilogue staging: 64 rows x 64 columns per wave and pass, 8 x 17,408 B

struct MxParams {
    const char* xh;
    const char* xl4;
    const char* x4;
    const char* xs;
    const int32_t* lens;
    const char* wh;          // [N-tile][K-step (padded to a multiple of 4)][16 KiB]: LDS images of w_h
    const char* wq;          // [N-tile][super-step][MX_WQ_BLOCK]: LDS images of w_4 | w_l6 | scales
    const float* bias;
    const float* scale;
    const float* shift;
    char* yh;                // output planes (chunk-major), or
    char* yl4;
    char* y4;
    char* ys;
    float* yf;               // ... fp32 row-major (B, T, ldy)
    int64_t ldy;
    int64_t T;               // rows per (utterance, chunk) of the input planes
    int64_t Tout;            // ... of the output (planes and fp32 rows): ktf_tdnn_out_len(T); == T with SAME padding and no subsampling
    int32_t units, nch_in, nctx, nk, nss, nch_out, stat_slots;
    int32_t sub, start, cut; // output row t reads the input rows start + t * sub + ctx[k] (clamped to the utterance); an utterance of len
                             // rows has ceil((len - cut - start) / sub) output rows (tdnn.py:224-249: VALID padding, subsampling_factor)
    unsigned long long ctx_pk[2];      // the (sorted) context offsets as signed bytes, offset k in byte k
    const int32_t* row_starts;         // ktf_tdnn_mx_flat: (B + 1) prefix sums of the lengths and ktf_flat_row_map's table (flat row tiles), else NULL
    const int32_t* row_map;
    uint32_t t_div_m, t_div_s;         // ... x / T as __umulhi(x, t_div_m) >> t_div_s for x < 2^31 (t_div_m == 0: T == 1): the row lookups of a batch
                                       // whose every utterance has all T rows (row_starts[B] == B * T) are arithmetic, no table load
};

// raw buffer resource over [ptr, ptr + 4 GiB): stride 0, no range limit below 2^32 - 1, gfx950's dword-3 (32-bit data format)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mx_rsrc(const void* ptr, int bytes = -1) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);      // (bytes: offsets at or beyond it read as 0)
}
// ... the same four words assembled by hand and passed through an empty asm: FOUR scalar registers the compiler has to keep as they are. A
// resource made by the builtin is a recipe to it: around every use it re-assembled the constant upper half (two s_mov per DMA) and lent the
// registers out in between.
typedef int mx_rs4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mx_rsrc_pinned(const void* ptr, int bytes = -1) {
    const unsigned long long a = (unsigned long long)ptr;
    mx_rs4 w{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), bytes, 0x00020000};
    asm volatile("" : "+s"(w));
    __amdgpu_buffer_rsrc_t r;
    __builtin_memcpy(&r, &w, 16);
    return r;
}

// E8M0 scale byte of an e2m1 block whose largest magnitude is m: the maximum lands in the top binade [4, 8) x scale, one
// binade lower when it would round past 6 (mantissa >= 1.75)
__device__ __forceinline__ unsigned mx_fp4_scale_byte(float m) {
    const unsigned bits = __float_as_uint(m);
    int byte = (int)(bits >> 23) - 2 + ((bits & 0x7fffffu) >= 0x600000u ? 1 : 0);
    return (unsigned)(byte < 1 ? 1 : byte);
}


// unit (inside the 256-unit tile) that column `m` of unit block `cb` of the weight images holds: inside each 32-unit chunk the
// order is permuted so that, with the weights as the A operand, lane quarter q4 of a frame owns units 8 q4 .. 8 q4 + 7 of the chunk
__device__ __forceinline__ int mx_unit(int cb, int m) { return (cb >> 1) * 32 + (m >> 2) * 8 + (cb & 1) * 4 + (m & 3); }

// registers (a, b, c, d) of the four lanes of a frame (lane quarter q = 0..3) hold element [register][q]: afterwards lane quarter q
// holds elements [q][0..3] (a 4 x 4 transpose between register index and lane quarter)
__device__ __forceinline__ void mx_transpose4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1];
    r = __builtin_amdgcn_permlane16_swap(c, d, false, false); c = r[0]; d = r[1];
    r = __builtin_amdgcn_permlane32_swap(a, c, false, false); a = r[0]; c = r[1];
    r = __builtin_amdgcn_permlane32_swap(b, d, false, false); b = r[0]; d = r[1];
}

// Plane encoder. Eight consecutive values of a 32-value block (this lane's quarter; the other three quarters sit in lanes ^ 16, ^ 32,
// ^ 48), each within [-65504, 65504] -> this lane's 16 bytes of the half piece, its dword of the two e2m1 records, the block's scale
// word. Same arithmetic as mx_encode32 (tdnn_mx.hip, ktf_mx_planes) and mx.encode_activations. The work is one long dependency chain (values ->
// half -> residual -> maxima -> two cross-lane steps -> scale -> conversions) and a wave has at most one partner on its SIMD, so
// TWO blocks (N = 2: two row blocks of the tile) are encoded with their chains interleaved statement by statement: alone a block
// took ~600 clk for ~70 instructions.
template <int N>
__device__ __forceinline__ void mx_encode8(const float (&v)[N][8], u32x4 (&hp)[N], unsigned (&l4)[N], unsigned (&h4)[N], unsigned (&sw)[N]) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    float lo[N][8];
    unsigned hw[N][4];
    float mv[N], ml[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { mv[n] = 0.0f; ml[n] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float a = v[n][2 * k], b = v[n][2 * k + 1];
            const h2 hh = __builtin_convertvector(f2{a, b}, h2);
            hw[n][k] = __builtin_bit_cast(unsigned, hh);
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[n][2 * k]) : "v"(hw[n][k]), "v"(a));
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[n][2 * k + 1]) : "v"(hw[n][k]), "v"(b));
            asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(mv[n]) : "v"(a), "v"(b), "v"(mv[n]));
            asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(ml[n]) : "v"(lo[n][2 * k]), "v"(lo[n][2 * k + 1]), "v"(ml[n]));
        }
    }
    // the maxima over the four lanes of a frame: all 2 N values per cross-lane step
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        unsigned o[N][2][2];
#pragma unroll
        for (int n = 0; n < N; ++n) {
            if (step == 0) {
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(mv[n]), __float_as_uint(mv[n]), false, false);
                o[n][0][0] = r[0]; o[n][0][1] = r[1];
                r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ml[n]), __float_as_uint(ml[n]), false, false);
                o[n][1][0] = r[0]; o[n][1][1] = r[1];
            } else {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mv[n]), __float_as_uint(mv[n]), false, false);
                o[n][0][0] = r[0]; o[n][0][1] = r[1];
                r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ml[n]), __float_as_uint(ml[n]), false, false);
                o[n][1][0] = r[0]; o[n][1][1] = r[1];
            }
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {      // (magnitudes: plain v_max_f32 -- from C the compiler quiets both operands of every fmaxf first)
            asm("v_max_f32 %0, %1, %2" : "=v"(mv[n]) : "v"(o[n][0][0]), "v"(o[n][0][1]));
            asm("v_max_f32 %0, %1, %2" : "=v"(ml[n]) : "v"(o[n][1][0]), "v"(o[n][1][1]));
        }
    }
    float sh[N], sl[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float mh = (float)(_Float16)mv[n];
        const unsigned bh = mx_fp4_scale_byte(mh), bl = mx_fp4_scale_byte(ml[n]);
        sh[n] = __uint_as_float(bh << 23);
        sl[n] = __uint_as_float(bl << 23);
        sw[n] = bl | (bh << 8);
    }
    // every conversion writes one byte of a register of its own (a chain of four through one register runs at the instruction's latency)
    unsigned xb[N][4], yb[N][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            xb[n][k] = 0;
            yb[n][k] = 0;
        }
    }
#define MX_ENC8_S(s_)                                                                                                   \
    _Pragma("unroll") for (int n = 0; n < N; ++n) {                                                                    \
        xb[n][s_] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(xb[n][s_], lo[n][2 * s_], lo[n][2 * s_ + 1], sl[n], s_);  \
        yb[n][s_] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(yb[n][s_], __builtin_bit_cast(h2, hw[n][s_]), sh[n], s_); \
    }
    MX_ENC8_S(0) MX_ENC8_S(1) MX_ENC8_S(2) MX_ENC8_S(3)
#undef MX_ENC8_S
#pragma unroll
    for (int n = 0; n < N; ++n) {
        l4[n] = (xb[n][0] | xb[n][1]) | (xb[n][2] | xb[n][3]);
        h4[n] = (yb[n][0] | yb[n][1]) | (yb[n][2] | yb[n][3]);
        hp[n] = u32x4{hw[n][0], hw[n][1], hw[n][2], hw[n][3]};
    }
}

#define MX_OUT_PLANES 0
#define MX_OUT_F32 1
#define MX_OUT_STATS 2

// 32 values, each within [-65504, 65504] (the callers clamp: half planes saturate instead of overflowing to inf) -> half plane
// piece (64 B), e2m1 images of the residual and of the half value (16 B each), scale word.
// Per pair of values: v_cvt_pk_f16_f32, two v_fma_mix_f32 (residual = value - half, exact), two v_max3_f32 (the maxima of the
// magnitudes: rounding to half is monotonic, so the largest half magnitude is the half of the largest magnitude), then
// v_cvt_scalef32_pk_fp4_{f32,f16}: ~135 vector instructions per 32 values (the plain-C form was ~430).
typedef __attribute__((ext_vector_type(2))) float mx_f2;
typedef __attribute__((ext_vector_type(2))) _Float16 mx_h2;
__device__ __forceinline__ void mx_encode32(const float (&v)[32], u32x4 (&hp)[4], u32x4& l4, u32x4& h4, unsigned& sw) {
    float lo[32];
    unsigned hw[16];
    float mv = 0.0f, ml = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float a = v[2 * k], b = v[2 * k + 1];
        const mx_h2 hh = __builtin_convertvector(mx_f2{a, b}, mx_h2);
        hw[k] = __builtin_bit_cast(unsigned, hh);
        // (spelled in assembly: from C the compiler emits v_cvt_f32_f16 + v_sub_f32 and quiets every v_max operand first)
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[2 * k]) : "v"(hw[k]), "v"(a));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[2 * k + 1]) : "v"(hw[k]), "v"(b));
        asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(mv) : "v"(a), "v"(b), "v"(mv));
        asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(ml) : "v"(lo[2 * k]), "v"(lo[2 * k + 1]), "v"(ml));
    }
    const float mh = (float)(_Float16)mv;
    const unsigned bh = mx_fp4_scale_byte(mh), bl = mx_fp4_scale_byte(ml);
    const float sh = __uint_as_float(bh << 23), sl = __uint_as_float(bl << 23);
    unsigned wl[4], wh_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {            // dword k = elements 8k .. 8k+7, byte s = elements 8k+2s, 8k+2s+1
        unsigned x = 0, y = 0;
#define MX_ENC_S(s_)                                                                                                   \
        x = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(x, lo[8 * k + 2 * s_], lo[8 * k + 2 * s_ + 1], sl, s_);           \
        y = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(y, __builtin_bit_cast(mx_h2, hw[4 * k + s_]), sh, s_);
        MX_ENC_S(0) MX_ENC_S(1) MX_ENC_S(2) MX_ENC_S(3)
#undef MX_ENC_S
        wl[k] = x;
        wh_[k] = y;
        hp[k] = u32x4{hw[4 * k], hw[4 * k + 1], hw[4 * k + 2], hw[4 * k + 3]};
    }
    l4 = u32x4{wl[0], wl[1], wl[2], wl[3]};
    h4 = u32x4{wh_[0], wh_[1], wh_[2], wh_[3]};
    sw = bl | (bh << 8);
}

// ------------------------------------------------------------------------------------ shared by the epilogues (tdnn_mx_epilogue.inc)
__device__ __forceinline__ float mx_act(float v, int act) { return act == KTF_ACT_RELU ? fmaxf(v, 0.0f) : v; }

__device__ __forceinline__ void mx_stats_out(double* __restrict__ stats, const MxParams& p, int b, int slot, int n, double s, double q) {
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

// tdnn_mxl.hip: the loader-wave kernel (include/ktf_hip.h, KTF_TDNN_MX_LOADER); `p` as filled by mx_launch
int mxl_launch(const MxParams& p, int64_t B, int act, int out_kind, double* stats, hipStream_t st);
