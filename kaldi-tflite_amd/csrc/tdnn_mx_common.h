// Shared by the two KTF_GEMM_F16MX kernels (tdnn_mx.hip: 256 x 256 tile on eight waves; tdnn_mxl.hip: 192 x 256 tile on eight matrix
// waves + four loader waves): operand types, the parameter block, the E8M0 scale rule.
#pragma once
#include "tdnn_common.h"
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
#define MX_EPI_PITCH 260

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
    int64_t T;
    int32_t units, nch_in, nctx, nk, nss, nch_out, stat_slots;
    unsigned long long ctx_pk[2];      // the (sorted) context offsets as signed bytes, offset k in byte k
};

// E8M0 scale byte of an e2m1 block whose largest magnitude is m: the maximum lands in the top binade [4, 8) x scale, one
// binade lower when it would round past 6 (mantissa >= 1.75)
__device__ __forceinline__ unsigned mx_fp4_scale_byte(float m) {
    const unsigned bits = __float_as_uint(m);
    int byte = (int)(bits >> 23) - 2 + ((bits & 0x7fffffu) >= 0x600000u ? 1 : 0);
    return (unsigned)(byte < 1 ? 1 : byte);
}


#define MX_OUT_PLANES 0
#define MX_OUT_F32 1
#define MX_OUT_STATS 2

// tdnn_mxl.hip: the loader-wave kernel (include/ktf_hip.h, KTF_TDNN_MX_LOADER); `p` as filled by mx_launch
int mxl_launch(const MxParams& p, int64_t B, int act, int out_kind, double* stats, hipStream_t st);
