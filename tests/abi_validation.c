// Host-side sanitizer run of the C-ABI's argument validation (no GPU needed: every call below must be rejected before any
// HIP call). Build + run: tools/asan_abi.sh
#include <stdio.h>
#include <string.h>
#include "ktf_hip.h"
static int fails = 0;
#define EXPECT_EINVAL(call)                                                       \
    do {                                                                          \
        int rc_ = (call);                                                         \
        char buf_[256];                                                           \
        ktf_last_error(buf_, sizeof buf_);                                        \
        if (rc_ != KTF_EINVAL) { printf("FAIL %s -> %d\n", #call, rc_); ++fails; } \
        else printf("ok   %-28.28s : %s\n", #call, buf_);                         \
    } while (0)
int main(void) {
    float f[64] = {0};
    double d[64] = {0};
    int32_t l[4] = {1, 1, 1, 1};
    uint32_t u[4] = {0};
    KtfTdnnDesc t;
    memset(&t, 0, sizeof t);
    t.units = 8; t.din = 8; t.din_pad = 32; t.nctx = 1; t.subsampling = 1; t.gemm = KTF_GEMM_F32;
    EXPECT_EINVAL(ktf_tdnn(NULL, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    t.nctx = 17;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    t.nctx = 2; t.ctx[0] = 1; t.ctx[1] = 0;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    t.nctx = 1; t.ctx[0] = 0; t.din_pad = 30;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    t.din_pad = 32; t.subsampling = 0;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    EXPECT_EINVAL(ktf_tdnn(NULL, 1, 0, 32, NULL, &t, f, NULL, NULL, NULL, NULL, NULL, 8, NULL, NULL));  /* ... also when the input is EMPTY (T == 0): the contract does not depend on the data */
    t.subsampling = 1;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, f, NULL, f, 8, NULL, NULL));       /* scale without shift */
    EXPECT_EINVAL(ktf_tdnn(NULL, 1, 0, 32, NULL, &t, f, NULL, NULL, f, NULL, NULL, 8, NULL, NULL)); /* ... on an empty input too */
    EXPECT_EINVAL(ktf_tdnn(NULL, 1, 0, 32, NULL, &t, NULL, NULL, NULL, NULL, NULL, NULL, 8, NULL, NULL)); /* the weights are never optional */
    EXPECT_EINVAL(ktf_tdnn(f, -1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    t.gemm = 77;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    t.gemm = KTF_GEMM_BF16; t.act = KTF_ACT_SELU; t.x_dtype = KTF_BF16; t.w_dtype = KTF_BF16; t.y_dtype = KTF_BF16;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));       /* an unfused activation outside fp32 */
    t.gemm = KTF_GEMM_F32; t.act = KTF_ACT_SOFTMAX + 1; t.x_dtype = t.w_dtype = t.y_dtype = KTF_F32;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));
    t.act = KTF_ACT_NONE;
    t.gemm = KTF_GEMM_BF16X4;                                                                          /* pairs in, pairs or fp32 out */
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));       /* fp32 operands */
    t.x_dtype = t.w_dtype = KTF_BF16P; t.y_dtype = KTF_BF16;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));       /* 16-bit output */
    t.y_dtype = KTF_F32;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, f, NULL, NULL, NULL, f, 8, NULL, NULL));          /* a residual plane */
    t.flags = KTF_TDNN_W_TILED;
    EXPECT_EINVAL(ktf_tdnn_stats(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, d, NULL));          /* no flag but KTF_TDNN_DET_STATS */
    t.flags = 0;
    t.gemm = KTF_GEMM_BF16; t.x_dtype = t.w_dtype = KTF_BF16; t.y_dtype = KTF_BF16P;
    EXPECT_EINVAL(ktf_tdnn(f, 1, 1, 32, NULL, &t, f, NULL, NULL, NULL, NULL, f, 8, NULL, NULL));       /* pairs from a 16-bit kernel */
    t.gemm = KTF_GEMM_F32; t.x_dtype = t.w_dtype = t.y_dtype = KTF_F32;
    t.gemm = KTF_GEMM_F16MX;
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, NULL, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL, NULL, NULL, f, 8, NULL));
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 8, NULL));     /* no output */
    t.valid = 1;                                                                                           /* VALID padding / subsampling: */
    EXPECT_EINVAL(ktf_tdnn_mx_stats(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, d, NULL));          /* ... not with the fused pooling */
    t.flags = KTF_TDNN_MX_LOADER;
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL, NULL, NULL, f, 8, NULL));     /* ... not on the loader-wave kernel */
    t.flags = 0; t.valid = 0; t.subsampling = 0;
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL, NULL, NULL, f, 8, NULL));
    EXPECT_EINVAL(ktf_tdnn_out_lens(NULL, 1, &t, NULL, NULL));
    t.subsampling = 1; t.ctx[0] = 300;
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL, NULL, NULL, f, 8, NULL));
    EXPECT_EINVAL(ktf_tdnn_mx_stats(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL));
    t.ctx[0] = 0; t.flags = KTF_TDNN_MX_LOADER;                                                            /* the loader-wave kernel ... */
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, f, f, f, f, f, f, NULL, 0, NULL));     /* ... writes planes without scale / shift */
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, f, 70000, 40000, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL, NULL, NULL, f, 8, NULL)); /* ... 32-bit row space */
    EXPECT_EINVAL(ktf_stats_finalize_slots(d, 4, 0, NULL, 500, 1, 4, 1, 1e-10f, f, 8, NULL));             /* slot_rows */
    if (ktf_mx_slot_rows(KTF_TDNN_MX_LOADER) != 96 || ktf_mx_slot_rows(0) != 128 || ktf_mx_stats_slots(998, KTF_TDNN_MX_LOADER) != 12 ||
        ktf_mx_stats_slots(998, 0) != ktf_stats_slots(998)) { printf("FAIL: ktf_mx_slot_rows / ktf_mx_stats_slots\n"); ++fails; }
    t.flags = 0;
    EXPECT_EINVAL(ktf_mx_planes(NULL, 1, 1, 8, 8, NULL, f, f, f, f, NULL));
    EXPECT_EINVAL(ktf_mx_planes(f, 1, 1, 8, 4, NULL, f, f, f, f, NULL));
    EXPECT_EINVAL(ktf_xvec_tail_f32(f, 8, d, 0, 128, l, 1, 1, 4, 1, 1e-10f, f, 8, NULL, 8, NULL, f, NULL, 4, f, u, f, NULL, 1, 0, NULL));   /* pooled and sums */
    EXPECT_EINVAL(ktf_xvec_tail_f32(f, 8, NULL, 0, 128, l, 1, 1, 4, 1, 1e-10f, f, 6, NULL, 8, NULL, f, NULL, 4, f, u, f, NULL, 1, 0, NULL));  /* ldw */
    EXPECT_EINVAL(ktf_xvec_tail_f32(f, 8, NULL, 0, 128, l, 1, 1, 4000, 1, 1e-10f, f, 8000, NULL, 8, NULL, f, NULL, 4, f, u, f, NULL, 1, 0, NULL));
    EXPECT_EINVAL(ktf_xvec_tail_f32(f, 8, NULL, 0, 128, NULL, 1, 1, 4, 1, 1e-10f, f, 8, NULL, 8, NULL, f, NULL, 4, f, u, f, NULL, 1, KTF_TAIL_SKIP_EMPTY, NULL));   /* skip without lens */
    EXPECT_EINVAL(ktf_xvec_post_f32(NULL, 1, 8, 4, NULL, f, NULL, f, NULL));
    EXPECT_EINVAL(ktf_stats_finalize(NULL, NULL, 1, 1, 4, 1, 1e-10f, f, 8, NULL));
    EXPECT_EINVAL(ktf_stats_finalize_slots(d, 1, 128, NULL, 1000, 1, 4, 1, 1e-10f, f, 8, NULL));       /* too few slots */
    EXPECT_EINVAL(ktf_route_short(NULL, 1, 400, l, l, NULL, 0, NULL));
    t.gemm = KTF_GEMM_BF16X3; t.x_dtype = KTF_BF16; t.w_dtype = KTF_BF16; t.y_dtype = KTF_F32; t.units = 256;
    EXPECT_EINVAL(ktf_tdnn_split_flat(f, f, 1, 1, 32, NULL, NULL, &t, f, f, NULL, NULL, NULL, f, NULL, 256, NULL));        /* no row map */
    t.valid = 1;
    EXPECT_EINVAL(ktf_tdnn_split_flat(f, f, 1, 1, 32, l, NULL, &t, f, f, NULL, NULL, NULL, f, NULL, 256, NULL));           /* VALID padding */
    t.valid = 0; t.act = KTF_ACT_TANH;
    EXPECT_EINVAL(ktf_tdnn_split_flat(f, f, 1, 1, 32, l, NULL, &t, f, f, NULL, NULL, NULL, f, NULL, 256, NULL));           /* fuses ReLU / none */
    t.act = KTF_ACT_NONE;
    EXPECT_EINVAL(ktf_tdnn_split_flat(f, f, 5000, 1, 32, l, NULL, &t, f, f, NULL, NULL, NULL, f, NULL, 256, NULL));        /* B > 4095 */
    EXPECT_EINVAL(ktf_tdnn_split_flat_stats(f, f, 1, 1, 32, NULL, NULL, &t, f, f, NULL, NULL, NULL, d, NULL));             /* no row map */
    EXPECT_EINVAL(ktf_tdnn_split_flat_stats(f, f, 1, 1, 32, l, NULL, &t, f, f, NULL, NULL, NULL, NULL, NULL));             /* no sums */
    t.valid = 1;
    EXPECT_EINVAL(ktf_tdnn_split_flat_stats(f, f, 1, 1, 32, l, NULL, &t, f, f, NULL, NULL, NULL, d, NULL));                /* VALID padding */
    t.valid = 0; t.flags = 16;
    EXPECT_EINVAL(ktf_tdnn_split_flat_stats(f, f, 1, 1, 32, l, NULL, &t, f, f, NULL, NULL, NULL, d, NULL));                /* a flag that left the interface */
    t.flags = 0; t.gemm = 4;
    EXPECT_EINVAL(ktf_tdnn_split(f, f, 1, 1, 32, NULL, &t, f, f, NULL, NULL, NULL, f, NULL, 256, NULL, NULL));       /* a mode that left the interface */
    t.gemm = KTF_GEMM_BF16X3;
    EXPECT_EINVAL(ktf_stats_finalize_flat(d, 1, l, 1000, 1, 4, 1, 1e-10f, f, 8, NULL));                              /* too few slots */
    EXPECT_EINVAL(ktf_flat_row_map(NULL, 4, 100, (int32_t*)l, NULL));                                                 /* no prefix sums */
    EXPECT_EINVAL(ktf_flat_row_map(l, 5000, 100, (int32_t*)l, NULL));                                                /* B > 4095 */
    EXPECT_EINVAL(ktf_stats_finalize_flat(d, 9, NULL, 1000, 1, 4, 1, 1e-10f, f, 8, NULL));                           /* no row map */
    t.gemm = KTF_GEMM_F16MX; t.units = 256; t.din = 32; t.din_pad = 32;
    EXPECT_EINVAL(ktf_tdnn_mx_flat(f, f, f, f, 1, 1, NULL, (int32_t*)l, &t, f, f, NULL, NULL, NULL, f, f, f, f, NULL));           /* no prefix sums */
    EXPECT_EINVAL(ktf_tdnn_mx_flat(f, f, f, f, 1, 1, l, NULL, &t, f, f, NULL, NULL, NULL, f, f, f, f, NULL));                     /* no row table */
    EXPECT_EINVAL(ktf_tdnn_mx_flat(f, f, f, f, 5000, 1, l, (int32_t*)l, &t, f, f, NULL, NULL, NULL, f, f, f, f, NULL));           /* B > 4095 */
    t.subsampling = 2;
    EXPECT_EINVAL(ktf_tdnn_mx_flat(f, f, f, f, 1, 8, l, (int32_t*)l, &t, f, f, NULL, NULL, NULL, f, f, f, f, NULL));              /* subsampling */
    t.subsampling = 1;
    EXPECT_EINVAL(ktf_tdnn_mx_flat_stats(f, f, f, f, 1, 1, l, (int32_t*)l, &t, f, f, NULL, NULL, NULL, NULL, NULL));              /* no sums */
    t.din = t.din_pad = 32 * 1200;                                                                                                  /* 1200 K-steps: the tile's K-step table ... */
    EXPECT_EINVAL(ktf_tdnn_mx(f, f, f, f, 1, 1, NULL, &t, f, f, NULL, NULL, NULL, f, f, f, f, NULL, 0, NULL));                   /* ... would not fit the LDS (1144) */
    t.din = t.din_pad = 32;
    t.gemm = KTF_GEMM_F32; t.x_dtype = t.w_dtype = t.y_dtype = KTF_F32; t.units = 8;
    EXPECT_EINVAL(ktf_convert_pad(NULL, KTF_F32, 1, 4, 4, f, KTF_F32, 4, NULL));
    EXPECT_EINVAL(ktf_split_bf16(NULL, 1, 4, 4, f, f, 32, NULL));
    EXPECT_EINVAL(ktf_split_bf16_rows(NULL, 1, 1, 4, 4, NULL, f, f, 32, NULL));
    EXPECT_EINVAL(ktf_split_bf16_rows(f, 1, 1, 4, 2, NULL, f, f, 32, NULL));                             /* ld_src < D */
    EXPECT_EINVAL(ktf_affine_act_f32(f, 1, 4, 99, NULL, NULL, f, NULL));
    EXPECT_EINVAL(ktf_affine_act_f32(f, 1, 4, KTF_ACT_SOFTMAX, NULL, NULL, f, NULL));                     /* a row operation: ktf_activation_f32 */
    EXPECT_EINVAL(ktf_activation_f32(NULL, 1, 1, 4, 4, NULL, KTF_ACT_ELU, NULL, NULL, NULL));
    EXPECT_EINVAL(ktf_activation_f32(f, 1, 1, 8, 4, NULL, KTF_ACT_ELU, NULL, NULL, NULL));               /* ld < D */
    EXPECT_EINVAL(ktf_activation_f32(f, 1, 1, 4, 4, NULL, KTF_ACT_SOFTMAX + 1, NULL, NULL, NULL));
    EXPECT_EINVAL(ktf_activation_f32(f, 1, 1, 4, 4, NULL, KTF_ACT_GELU, f, NULL, NULL));                  /* scale without shift */
    char small[4];
    size_t n = ktf_last_error(small, sizeof small);      /* truncation must stay inside the buffer */
    printf("last_error length %zu, truncated copy '%s'\n", n, small);
    printf(fails ? "FAILED %d\n" : "all rejected as KTF_EINVAL\n", fails);
    return fails != 0;
}
