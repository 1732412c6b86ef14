// C-ABI of the TDNN layer (include/ktf_hip.h: ktf_tdnn, ktf_tdnn_stats, ktf_tdnn_split, ktf_tdnn_split_stats) -- argument
// validation, the kernel parameter block, and the dispatch to the kernel families (tdnn_f32.hip, tdnn_bf16.hip, tdnn_split.hip;
// KTF_GEMM_F16MX has its own entry points in tdnn_mx.hip) -- plus the small elementwise / conversion / pooling-finalize kernels
// of the layer stack. ktf_tdnn_last_kernel() names the kernel family the calling thread's last ktf_tdnn* call launched
// (tests/test_gpu_dispatch.py pins the (mode, shape) -> kernel map with it).
#include "tdnn_common.h"

// ------------------------------------------------------------------------------------ elementwise helpers
__global__ void affine_act_kernel(const float* __restrict__ x, int64_t total, int D, int act,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  float* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        float v = apply_act_ext(x[e], act);
        if (scale) v *= scale[d];
        if (shift) v += shift[d];
        y[e] = v;
    }
}

// One wave per output row, in place: y = act(y) * scale + shift over the rows a TDNN launch wrote (out rows of utterance b =
// ceil((lens[b] - trim) / sub), trim / sub from the layer's padding and subsampling; the stand-alone entry point passes 0 / 1).
__global__ __launch_bounds__(256) void act_rows_kernel(float* __restrict__ y, int64_t B, int64_t T, int D, int64_t ld,
                                                      const int32_t* __restrict__ lens, int64_t Tin, int trim, int sub, int act,
                                                      const float* __restrict__ scale, const float* __restrict__ shift) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int64_t b = row / T, t = row - b * T;
    const int64_t n = (lens ? (int64_t)lens[b] : Tin) - trim;
    const int64_t rows = n <= 0 ? 0 : (n + sub - 1) / sub;
    if (t >= rows) return;
    float* r = y + row * ld;
    float inv = 1.0f, mx = 0.0f;
    if (act == KTF_ACT_SOFTMAX) {
        mx = -INFINITY;
        for (int d = lane; d < D; d += 64) mx = fmaxf(mx, r[d]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float s = 0.0f;
        for (int d = lane; d < D; d += 64) s += expf(r[d] - mx);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        inv = 1.0f / s;
    }
    for (int d = lane; d < D; d += 64) {
        float v = act == KTF_ACT_SOFTMAX ? expf(r[d] - mx) * inv : apply_act_ext(r[d], act);
        if (scale) v = v * scale[d] + shift[d];
        r[d] = v;
    }
}

static int act_rows_launch(float* y, int64_t B, int64_t T, int32_t D, int64_t ld, const int32_t* lens, int64_t Tin, int trim, int sub,
                           int act, const float* scale, const float* shift, hipStream_t st) {
    const int64_t rows = B * T;
    if (rows == 0) return KTF_OK;
    KTF_REQUIRE((rows + 3) / 4 < (1ll << 31), "activation pass: too many rows");
    hipLaunchKernelGGL(act_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, y, B, T, D, ld, lens, Tin, trim, sub, act, scale, shift);
    return KTF_OK;
}

extern "C" int ktf_activation_f32(float* y, int64_t B, int64_t T, int32_t D, int64_t ld, const int32_t* lens, int32_t act,
                                  const float* scale, const float* shift, void* stream) {
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0 && ld >= D, "ktf_activation_f32: bad sizes");
    KTF_REQUIRE(y || B * T == 0, "ktf_activation_f32: null argument");
    KTF_REQUIRE(act >= KTF_ACT_NONE && act <= KTF_ACT_SOFTMAX, "ktf_activation_f32: bad activation %d", act);
    KTF_REQUIRE((scale == nullptr) == (shift == nullptr), "ktf_activation_f32: scale and shift go together");
    const int rc = act_rows_launch(y, B, T, D, ld, lens, T, 0, 1, act, scale, shift, (hipStream_t)stream);
    if (rc != KTF_OK) return rc;
    KTF_CHECK_LAUNCH("ktf_activation_f32");
    return KTF_OK;
}

template <typename S>
__device__ __forceinline__ float cp_load(const S* p);
template <> __device__ __forceinline__ float cp_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float cp_load<unsigned short>(const unsigned short* p) { return bf2f(*p); }
template <typename Dd>
__device__ __forceinline__ void cp_store(Dd* p, float v);
template <> __device__ __forceinline__ void cp_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void cp_store<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }

template <typename S, typename Dd>
__global__ void convert_pad_kernel(const S* __restrict__ src, int64_t rows, int D, int64_t lds_, Dd* __restrict__ dst,
                                   int64_t ldd) {
    const int64_t total = rows * ldd;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / ldd;
        const int d = (int)(e - r * ldd);
        cp_store<Dd>(dst + e, d < D ? cp_load<S>(src + r * lds_ + d) : 0.0f);
    }
}

thread_local const char* g_ktf_last_kernel = "";
extern "C" const char* ktf_tdnn_last_kernel(void) { return g_ktf_last_kernel; }

extern "C" int64_t ktf_tdnn_out_len(int64_t len, const KtfTdnnDesc* d) {
    if (!d || d->nctx <= 0 || d->nctx > 16 || d->subsampling <= 0) return -1;      // (ctx[] holds 16 offsets)
    int64_t start = 0, end = len;
    if (d->valid) {
        if (d->ctx[0] < 0) start = -d->ctx[0];
        if (d->ctx[d->nctx - 1] > 0) end = len - d->ctx[d->nctx - 1];
    }
    const int64_t n = end - start;
    return n <= 0 ? 0 : (n + d->subsampling - 1) / d->subsampling;
}

// out_lens[b] = ktf_tdnn_out_len(lens[b]) on the device: the entry points on the MX planes (ktf_tdnn_mx) have no out_lens argument
__global__ void tdnn_out_lens_kernel(const int32_t* __restrict__ lens, int64_t B, int start, int cut, int sub, int32_t* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        const int n = lens[b] - cut - start;
        out[b] = n <= 0 ? 0 : (n + sub - 1) / sub;
    }
}

extern "C" int ktf_tdnn_out_lens(const int32_t* lens, int64_t B, const KtfTdnnDesc* d, int32_t* out_lens, void* stream) {
    KTF_REQUIRE(lens && d && out_lens, "ktf_tdnn_out_lens: null argument");
    KTF_REQUIRE(B >= 0 && d->nctx >= 1 && d->nctx <= 16 && d->subsampling >= 1, "ktf_tdnn_out_lens: bad size / descriptor");
    if (B == 0) return KTF_OK;
    const int start = (d->valid && d->ctx[0] < 0) ? -d->ctx[0] : 0;
    const int cut = (d->valid && d->ctx[d->nctx - 1] > 0) ? d->ctx[d->nctx - 1] : 0;
    hipLaunchKernelGGL(tdnn_out_lens_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lens, B, start, cut,
                       d->subsampling, out_lens);
    KTF_CHECK_LAUNCH("ktf_tdnn_out_lens");
    return KTF_OK;
}

extern "C" int64_t ktf_flat_stats_slots(int64_t T) { return T <= 0 ? 1 : (T + 254) / 128; }

static int tdnn_launch(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
                       const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift,
                       void* y, int64_t ldy, int32_t* out_lens, double* stats_sums, void* stream,
                       const void* x_lo = nullptr, void* y_lo = nullptr, const int32_t* row_starts = nullptr, const int32_t* row_map = nullptr) {
    KTF_REQUIRE(d, "ktf_tdnn: null descriptor");
    // Everything that does not depend on the data is validated first -- descriptor, gemm / dtype combination, activation, scale / shift
    // pairing, sizes --, so that a malformed call is rejected whether or not its input happens to be empty; only the null-pointer checks
    // are relaxed for tensors without an element (an empty input: x, x_lo; no output row -- T == 0, or a VALID-padded layer's context
    // longer than the input --: y, y_lo).
    const bool no_in = B == 0 || T == 0;
    const bool split_in = d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16;     // activations as hi/lo bf16 planes
    KTF_REQUIRE(w && (no_in || x), "ktf_tdnn: null argument");
    if (split_in) KTF_REQUIRE(no_in || x_lo, "ktf_tdnn_split: null lo plane");
    if (y_lo) KTF_REQUIRE(split_in && d->y_dtype == KTF_BF16, "ktf_tdnn_split: a split output needs split input and y_dtype bf16");
    if (stats_sums && d->gemm == KTF_GEMM_BF16X4) {
        ldy = d->units;                                  // (any shape: the pair kernel pools the rows it would have written)
    } else if (stats_sums) {
        KTF_REQUIRE(((d->gemm == KTF_GEMM_BF16 && d->x_dtype == KTF_BF16) || (d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_F32) || split_in) &&
                        d->units > 128 && !d->valid && d->subsampling == 1,
                    "ktf_tdnn_stats: needs a ring kernel (bf16 or bf16x3 gemm, units > 128, SAME padding, no subsampling) or KTF_GEMM_BF16X4");
        ldy = (d->units + 3) / 4 * 4;
    }
    KTF_REQUIRE(B >= 0 && T >= 0, "ktf_tdnn: negative size");
    KTF_REQUIRE(!(d->flags & ~(KTF_TDNN_REF_TILES | KTF_TDNN_DET_STATS | KTF_TDNN_K_INTERLEAVED | KTF_TDNN_W_TILED | KTF_TDNN_MX_LOADER)),
                "ktf_tdnn: unknown bits in KtfTdnnDesc.flags (0x%x)", (unsigned)d->flags);
    KTF_REQUIRE(d->units > 0 && d->din > 0, "ktf_tdnn: units/din must be > 0");
    KTF_REQUIRE(d->nctx >= 1 && d->nctx <= 16, "ktf_tdnn: nctx %d outside [1,16]", d->nctx);
    for (int i = 1; i < d->nctx; ++i) KTF_REQUIRE(d->ctx[i] > d->ctx[i - 1], "ktf_tdnn: context must be strictly ascending");
    KTF_REQUIRE(d->subsampling > 0, "ktf_tdnn: subsampling_factor should be > 0");
    KTF_REQUIRE(d->din_pad >= d->din && d->din_pad % 32 == 0 && d->din_pad <= ldx, "ktf_tdnn: din_pad %d must be a multiple of 32 with din <= din_pad <= ldx", d->din_pad);
    KTF_REQUIRE(ldx % 8 == 0, "ktf_tdnn: ldx must be a multiple of 8");
    KTF_REQUIRE(ldy >= d->units, "ktf_tdnn: ldy < units");
    KTF_REQUIRE(d->act >= KTF_ACT_NONE && d->act <= KTF_ACT_SOFTMAX, "ktf_tdnn: bad activation %d", d->act);
    const bool act_pass = d->act > KTF_ACT_TANH;        // not fused by any epilogue: a second launch over the rows written
    if (act_pass) KTF_REQUIRE(d->gemm == KTF_GEMM_F32 && d->y_dtype == KTF_F32 && y && !stats_sums,
                              "ktf_tdnn: activation %d runs with KTF_GEMM_F32 and an fp32 output only (no fused pooling)", d->act);
    const bool y_pair = d->y_dtype == KTF_BF16P;        // pairs in fp32-sized slots: the fp32 and the pair kernels write them
    if (y_pair) KTF_REQUIRE((d->gemm == KTF_GEMM_F32 || d->gemm == KTF_GEMM_BF16X4) && (y || stats_sums) && !act_pass,
                            "ktf_tdnn: a KTF_BF16P output comes from KTF_GEMM_F32 or KTF_GEMM_BF16X4 (no fused pooling, fused activations only)");
    KTF_REQUIRE(y_pair || d->y_dtype == KTF_F32 || d->y_dtype == KTF_BF16, "ktf_tdnn: bad y_dtype");
    KTF_REQUIRE((scale == nullptr) == (shift == nullptr), "ktf_tdnn: scale and shift go together");
    KTF_REQUIRE(T < (1ll << 30) && B < 65536, "ktf_tdnn: T or B too large");
    const int64_t Tout = ktf_tdnn_out_len(T, d);
    if (B == 0) return KTF_OK;
    if (T == 0 || Tout <= 0) {                           // no output row: every length 0, nothing else is touched (callers of
        if (out_lens) (void)hipMemsetAsync(out_lens, 0, sizeof(int32_t) * B, (hipStream_t)stream);      // ktf_tdnn_stats zero the sums)
        return KTF_OK;
    }
    KTF_REQUIRE(y || stats_sums, "ktf_tdnn: null output");
    TdnnParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.lens = lens; p.w = w; p.w_lo = w_lo; p.bias = bias; p.scale = scale; p.shift = shift; p.y = y;
    p.x_lo = x_lo; p.y_lo = y_lo; p.row_starts = row_starts; p.row_map = row_map;
    p.out_lens = out_lens; p.T = T; p.ldx = ldx; p.ldy = ldy; p.Tout = Tout;
    p.units = d->units; p.din_pad = d->din_pad; p.nctx = d->nctx; p.sub = d->subsampling; p.valid = d->valid;
    p.act = act_pass ? KTF_ACT_NONE : d->act; p.y_dtype = y_pair ? KTF_F32 : d->y_dtype; p.y_pair = y_pair ? 1 : 0; p.ktot = d->nctx * d->din_pad;
    if (act_pass) p.scale = p.shift = nullptr;          // (the BatchNorm affine follows the activation: applied by the pass)
    p.stat_slots = (stats_sums && (d->flags & KTF_TDNN_DET_STATS)) ? (int32_t)(row_starts ? ktf_flat_stats_slots(Tout) : ktf_stats_slots(Tout)) : 0;
    p.kinter = (d->flags & KTF_TDNN_K_INTERLEAVED) ? 1 : 0;
    p.wtiled = (d->flags & KTF_TDNN_W_TILED) ? 1 : 0;
    if (p.wtiled) KTF_REQUIRE((split_in && d->units > 128 && ldy % 4 == 0), "ktf_tdnn: KTF_TDNN_W_TILED is implemented by the split-plane kernel only");
    if (p.kinter) KTF_REQUIRE((split_in && d->units > 128 && ldy % 4 == 0), "ktf_tdnn: KTF_TDNN_K_INTERLEAVED is implemented by the split-plane kernel only (ktf_tdnn_split*, units > 128)");
    for (int i = 0; i < d->nctx; ++i) p.ctx[i] = d->ctx[i];
    hipStream_t st = (hipStream_t)stream;
    if (d->gemm == KTF_GEMM_F32) {
        KTF_REQUIRE(d->x_dtype == KTF_F32 && d->w_dtype == KTF_F32, "ktf_tdnn: F32 gemm needs fp32 x and w");
        const int rc = tdnn_launch_f32(p, d, B, Tout, st);
        if (rc != KTF_OK || !act_pass) return rc;
        const int trim = d->valid ? (d->ctx[0] < 0 ? -d->ctx[0] : 0) + (d->ctx[d->nctx - 1] > 0 ? d->ctx[d->nctx - 1] : 0) : 0;
        const int rc2 = act_rows_launch((float*)y, B, Tout, d->units, ldy, lens, T, trim, d->subsampling, d->act, scale, shift, st);
        if (rc2 != KTF_OK) return rc2;
        KTF_CHECK_LAUNCH("ktf_tdnn (activation pass)");
        return KTF_OK;
    }
    if (d->gemm == KTF_GEMM_BF16X4) {
        KTF_REQUIRE(d->x_dtype == KTF_BF16P && d->w_dtype == KTF_BF16P && !w_lo, "ktf_tdnn: KTF_GEMM_BF16X4 takes x and w of KTF_BF16P (w_lo NULL)");
        KTF_REQUIRE(stats_sums || y_pair || d->y_dtype == KTF_F32, "ktf_tdnn: KTF_GEMM_BF16X4 writes KTF_F32 or KTF_BF16P rows");
        KTF_REQUIRE(!(d->flags & ~KTF_TDNN_DET_STATS), "ktf_tdnn: KTF_GEMM_BF16X4 takes no KTF_TDNN_* flag but KTF_TDNN_DET_STATS");
        p.stat_slots = (stats_sums && (d->flags & KTF_TDNN_DET_STATS)) ? (int32_t)ktf_tdnn_stats_slots(Tout, d->gemm) : 0;
        return tdnn_launch_x4(p, d, B, Tout, stats_sums, st);
    }
    if (d->gemm == KTF_GEMM_BF16 || d->gemm == KTF_GEMM_BF16X3) {
        const bool x3 = d->gemm == KTF_GEMM_BF16X3;
        if (x3) KTF_REQUIRE((d->x_dtype == KTF_F32 || split_in) && w_lo, "ktf_tdnn: BF16X3 needs fp32 activations (or hi/lo planes) and w_lo");
        if (split_in) KTF_REQUIRE(d->units > 128 && ldy % 4 == 0, "ktf_tdnn_split: runs on the 256x256 kernel only (units > 128, ldy %% 4 == 0)");
        if (x3 && d->units > 128 && ldy % 4 == 0) {
            KTF_REQUIRE(d->w_dtype == KTF_BF16, "ktf_tdnn: bf16 gemm needs bf16 weights");
            KTF_REQUIRE(d->y_dtype == KTF_BF16 || d->y_dtype == KTF_F32, "ktf_tdnn: bf16 gemm writes bf16 or fp32");
            return tdnn_launch_split(p, d, B, Tout, ldy, split_in, stats_sums, st);
        }
        return tdnn_launch_16(p, d, B, Tout, ldy, stats_sums, st);
    }
    KTF_REQUIRE(false, "ktf_tdnn: unknown gemm mode %d", d->gemm);
    return KTF_OK;
}

extern "C" int ktf_tdnn(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
                        const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift,
                        void* y, int64_t ldy, int32_t* out_lens, void* stream) {
    KTF_REQUIRE(y, "ktf_tdnn: null output");
    return tdnn_launch(x, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, y, ldy, out_lens, nullptr, stream);
}

extern "C" int ktf_tdnn_stats(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
                              const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift,
                              double* sums, void* stream) {
    KTF_REQUIRE(sums, "ktf_tdnn_stats: null sums");
    return tdnn_launch(x, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, nullptr, 0, nullptr, sums, stream);
}

extern "C" int ktf_tdnn_split(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* lens,
                              const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                              const float* shift, void* y, void* y_lo, int64_t ldy, int32_t* out_lens, void* stream) {
    KTF_REQUIRE(d && (y || T == 0 || ktf_tdnn_out_len(T, d) == 0), "ktf_tdnn_split: null argument");
    KTF_REQUIRE(d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16, "ktf_tdnn_split: needs KTF_GEMM_BF16X3 with x_dtype KTF_BF16 (hi/lo planes)");
    return tdnn_launch(x_hi, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, y, ldy, out_lens, nullptr, stream, x_lo, y_lo);
}

// ktf_tdnn_split over the batch's valid rows laid end to end (short utterances: a 1.5 s window fills 0.58 of a 256-row tile)
// the row table of the flat tiles, made once per batch for all its layers: (output row b * T + t or -1, frame t, utterance length, utterance b)
// of flat row R, for R < round_up(B * T, 256)
__global__ __launch_bounds__(256) void flat_row_map_kernel(const int32_t* __restrict__ rs, int B, int T, i32x4* __restrict__ map, int64_t rows) {
    const int64_t R = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (R >= rows) return;
    i32x4 e = {-1, 0, 1, 0};
    if (R < rs[B]) {
        int lo = 0, hi = B - 1;                              // the last b with rs[b] <= R (empty utterances repeat a start: the last one wins)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rs[mid] <= R) lo = mid; else hi = mid - 1;
        }
        const int s0 = rs[lo], t = (int)R - s0;
        e = i32x4{lo * T + t, t, rs[lo + 1] - s0, lo};
    }
    map[R] = e;
}

extern "C" int64_t ktf_flat_row_map_rows(int64_t B, int64_t T) { return B <= 0 || T <= 0 ? 0 : (B * T + 255) / 256 * 256; }
extern "C" int ktf_flat_row_map(const int32_t* row_starts, int64_t B, int64_t T, int32_t* map, void* stream) {
    KTF_REQUIRE(B >= 0 && T >= 0 && B <= 4095 && B * T < (1ll << 31), "ktf_flat_row_map: bad sizes");
    const int64_t rows = ktf_flat_row_map_rows(B, T);
    if (rows == 0) return KTF_OK;
    KTF_REQUIRE(row_starts && map, "ktf_flat_row_map: null argument");
    hipLaunchKernelGGL(flat_row_map_kernel, dim3((unsigned)(rows / 256)), dim3(256), 0, (hipStream_t)stream, row_starts, (int)B, (int)T,
                       reinterpret_cast<i32x4*>(map), rows);
    KTF_CHECK_LAUNCH("ktf_flat_row_map");
    return KTF_OK;
}

extern "C" int ktf_tdnn_split_flat(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* row_starts, const int32_t* row_map,
                                   const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                                   const float* shift, void* y, void* y_lo, int64_t ldy, void* stream) {
    KTF_REQUIRE(d && w && w_lo && (B == 0 || T == 0 || (x_hi && x_lo && row_starts && y)), "ktf_tdnn_split_flat: null argument");      // (empty tensors: null pointers)
    KTF_REQUIRE(d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16, "ktf_tdnn_split_flat: needs KTF_GEMM_BF16X3 on bf16 hi / lo planes");
    return tdnn_launch(x_hi, B, T, ldx, nullptr, d, w, w_lo, bias, scale, shift, y, ldy, nullptr, nullptr, stream, x_lo, y_lo, row_starts, row_map);
}

// ... with the reducing StatsPooling fused (ktf_tdnn_split_stats on flat row tiles)
extern "C" int ktf_tdnn_split_flat_stats(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* row_starts,
                                         const int32_t* row_map, const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                                         const float* shift, double* sums, void* stream) {
    KTF_REQUIRE(d && w && w_lo && sums && (B == 0 || T == 0 || (x_hi && x_lo && row_starts)), "ktf_tdnn_split_flat_stats: null argument");
    KTF_REQUIRE(d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16, "ktf_tdnn_split_flat_stats: needs KTF_GEMM_BF16X3 on bf16 hi / lo planes");
    return tdnn_launch(x_hi, B, T, ldx, nullptr, d, w, w_lo, bias, scale, shift, nullptr, 0, nullptr, sums, stream, x_lo, nullptr, row_starts, row_map);
}

extern "C" int ktf_tdnn_split_stats(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx,
                                    const int32_t* lens, const KtfTdnnDesc* d, const void* w, const void* w_lo,
                                    const float* bias, const float* scale, const float* shift, double* sums, void* stream) {
    KTF_REQUIRE(sums && d, "ktf_tdnn_split_stats: null argument");
    KTF_REQUIRE(d->gemm == KTF_GEMM_BF16X3 && d->x_dtype == KTF_BF16, "ktf_tdnn_split_stats: needs KTF_GEMM_BF16X3 with x_dtype KTF_BF16 (hi/lo planes)");
    return tdnn_launch(x_hi, B, T, ldx, lens, d, w, w_lo, bias, scale, shift, nullptr, 0, nullptr, sums, stream, x_lo, nullptr);
}

// fp32 rows -> the two bf16 planes of the split representation (hi = bf16(v), lo = bf16(v - hi)); pad columns zero
__global__ void split_bf16_kernel(const float* __restrict__ src, int64_t rows, int D, int64_t lds_, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo, int64_t ldd) {
    const int64_t total = rows * ldd;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / ldd;
        const int d = (int)(e - r * ldd);
        const float v = d < D ? src[r * lds_ + d] : 0.0f;
        const unsigned short h = f2bf(v);
        hi[e] = h;
        lo[e] = f2bf(v - bf2f(h));
    }
}

extern "C" int ktf_split_bf16(const float* src, int64_t rows, int32_t D, int64_t ld_src, void* hi, void* lo, int64_t ld_dst,
                              void* stream) {
    KTF_REQUIRE(rows >= 0 && D > 0 && ld_src >= D && ld_dst >= D, "ktf_split_bf16: bad sizes");
    if (rows == 0) return KTF_OK;                        // (an empty tensor: null pointers)
    KTF_REQUIRE(src && hi && lo, "ktf_split_bf16: null argument");
    const int64_t total = rows * ld_dst;
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, rows, D, ld_src,
                       (unsigned short*)hi, (unsigned short*)lo, ld_dst);
    KTF_CHECK_LAUNCH("ktf_split_bf16");
    return KTF_OK;
}

// ... of the VALID rows of a ragged batch only (rows at and beyond lens[b] are never read: every consumer clamps its row reads to the
// utterance): the second pass of an f16mx model over its few short utterances used to split all B x T rows (64 us per 1024 x 998)
#define SPLIT_RB 32
__global__ __launch_bounds__(256) void split_bf16_rows_kernel(const float* __restrict__ src, int64_t T, int D, int64_t lds_, const int32_t* __restrict__ lens,
                                                              unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int64_t ldd) {
    const int64_t b = blockIdx.y, t0 = (int64_t)blockIdx.x * SPLIT_RB;
    const int64_t len = lens ? (int64_t)lens[b] : T;
    if (t0 >= len) return;
    const int64_t rows = len - t0 < SPLIT_RB ? len - t0 : SPLIT_RB, r0 = b * T + t0;
    for (int64_t e = threadIdx.x; e < rows * ldd; e += 256) {
        const int64_t r = r0 + e / ldd;
        const int d = (int)(e % ldd);
        const float v = d < D ? src[r * lds_ + d] : 0.0f;
        const unsigned short h = f2bf(v);
        hi[r * ldd + d] = h;
        lo[r * ldd + d] = f2bf(v - bf2f(h));
    }
}

extern "C" int ktf_split_bf16_rows(const float* src, int64_t B, int64_t T, int32_t D, int64_t ld_src, const int32_t* lens, void* hi, void* lo,
                                   int64_t ld_dst, void* stream) {
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0 && ld_src >= D && ld_dst >= D && B < 65536, "ktf_split_bf16_rows: bad sizes");
    if (B == 0 || T == 0) return KTF_OK;
    KTF_REQUIRE(src && hi && lo, "ktf_split_bf16_rows: null argument");
    hipLaunchKernelGGL(split_bf16_rows_kernel, dim3((unsigned)ktf_cdiv(T, SPLIT_RB), (unsigned)B), dim3(256), 0, (hipStream_t)stream, src, T, D, ld_src,
                       lens, (unsigned short*)hi, (unsigned short*)lo, ld_dst);
    KTF_CHECK_LAUNCH("ktf_split_bf16_rows");
    return KTF_OK;
}

// 128-row blocks, rounded up to whole 256-row tiles (a 256-row tile always writes both of its blocks)
extern "C" int64_t ktf_stats_slots(int64_t T) { return T <= 0 ? 2 : 2 * ((T + 255) / 256); }
// ... of ktf_tdnn_stats by GEMM mode: KTF_GEMM_BF16X4 pools per 64-row tile (csrc/tdnn_pair.hip), the others per 128-row block
extern "C" int32_t ktf_tdnn_slot_rows(int32_t gemm) { return gemm == KTF_GEMM_BF16X4 ? 64 : 128; }
extern "C" int64_t ktf_tdnn_stats_slots(int64_t T, int32_t gemm) {
    if (gemm != KTF_GEMM_BF16X4) return ktf_stats_slots(T);
    return T <= 0 ? 1 : (T + 63) / 64;
}

// mean / std from the fp64 column sums of ktf_tdnn_stats: out[b, c] = mean, out[b, D + c] = sqrt(relu(E[x^2]-mean^2)+eps), NaN kept
__global__ void stats_finalize_kernel(const double* __restrict__ sums, int64_t slots, int slot_rows, const int32_t* __restrict__ lens, int64_t T,
                                      int64_t B, int D, int include_std, float eps, float* __restrict__ out, int64_t ldo,
                                      const int32_t* __restrict__ row_starts = nullptr) {
    const int64_t total = B * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / D;
        const int c = (int)(e - b * D);
        const int rs0 = row_starts ? row_starts[b] : 0;
        const int len = row_starts ? row_starts[b + 1] - rs0 : (lens ? lens[b] : (int)T);
        const double n = (double)len;
        double s = 0.0, q = 0.0;
        if (slots == 0) {
            s = sums[(b * 2) * D + c];
            q = sums[(b * 2 + 1) * D + c];
        } else {
            // blocks holding valid rows, added in block order (flat row tiles: the 128-row blocks of the flat row space the utterance touches)
            const int used = row_starts ? (len > 0 ? ((rs0 + len - 1) >> 7) - (rs0 >> 7) + 1 : 0) : (len + slot_rows - 1) / slot_rows;
            for (int k = 0; k < used; ++k) {
                s += sums[((b * slots + k) * 2) * D + c];
                q += sums[((b * slots + k) * 2 + 1) * D + c];
            }
        }
        const double mean = s / n;
        out[b * ldo + c] = (float)mean;
        if (include_std) {
            // relu as tf.nn.relu (stats_pooling.py:238): a NaN variance -- an utterance without a frame (0 / 0), NaN activations -- stays NaN;
            // fmax(NaN, 0) = 0 made sqrt(eps) of it (tools/fuzz_models.py seed 9102: an utterance a VALID-padded layer leaves no frame of)
            const double var = q / n - mean * mean;
            out[b * ldo + D + c] = (float)sqrt((var < 0.0 ? 0.0 : var) + (double)eps);
        }
    }
}

extern "C" int ktf_stats_finalize(const double* sums, const int32_t* lens, int64_t T, int64_t B, int32_t D,
                                  int32_t include_std, float eps, float* out, int64_t ld_out, void* stream) {
    KTF_REQUIRE(sums && out, "ktf_stats_finalize: null argument");
    KTF_REQUIRE(B >= 0 && D > 0 && ld_out >= (include_std ? 2 : 1) * (int64_t)D, "ktf_stats_finalize: bad sizes");
    if (B == 0) return KTF_OK;
    int blocks = ktf_cdiv(B * D, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sums, (int64_t)0, 128, lens, T, B, D, include_std, eps, out, ld_out);
    KTF_CHECK_LAUNCH("ktf_stats_finalize");
    return KTF_OK;
}

extern "C" int ktf_stats_finalize_slots(const double* sums, int64_t slots, int32_t slot_rows, const int32_t* lens, int64_t T, int64_t B, int32_t D,
                                        int32_t include_std, float eps, float* out, int64_t ld_out, void* stream) {
    KTF_REQUIRE(sums && out, "ktf_stats_finalize_slots: null argument");
    KTF_REQUIRE(B >= 0 && D > 0 && ld_out >= (include_std ? 2 : 1) * (int64_t)D, "ktf_stats_finalize_slots: bad sizes");
    KTF_REQUIRE(slot_rows > 0 && slots * slot_rows >= T, "ktf_stats_finalize_slots: %lld slots of %d rows do not cover %lld rows", (long long)slots, slot_rows, (long long)T);
    if (B == 0) return KTF_OK;
    int blocks = ktf_cdiv(B * D, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sums, slots, (int)slot_rows, lens, T, B, D, include_std, eps, out, ld_out);
    KTF_CHECK_LAUNCH("ktf_stats_finalize_slots");
    return KTF_OK;
}

extern "C" int ktf_stats_finalize_flat(const double* sums, int64_t slots, const int32_t* row_starts, int64_t T, int64_t B, int32_t D,
                                       int32_t include_std, float eps, float* out, int64_t ld_out, void* stream) {
    KTF_REQUIRE(sums && out && row_starts, "ktf_stats_finalize_flat: null argument");
    KTF_REQUIRE(B >= 0 && D > 0 && ld_out >= (include_std ? 2 : 1) * (int64_t)D, "ktf_stats_finalize_flat: bad sizes");
    KTF_REQUIRE(slots == 0 || slots >= ktf_flat_stats_slots(T), "ktf_stats_finalize_flat: %lld slots, utterances of up to %lld rows touch %lld blocks",
                (long long)slots, (long long)T, (long long)ktf_flat_stats_slots(T));
    if (B == 0) return KTF_OK;
    int blocks = ktf_cdiv(B * D, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sums, slots, 128, (const int32_t*)nullptr, T, B, D, include_std, eps,
                       out, ld_out, row_starts);
    KTF_CHECK_LAUNCH("ktf_stats_finalize_flat");
    return KTF_OK;
}

extern "C" int ktf_affine_act_f32(const float* x, int64_t rows, int32_t D, int32_t act, const float* scale,
                                  const float* shift, float* y, void* stream) {
    KTF_REQUIRE(rows >= 0 && D > 0, "ktf_affine_act_f32: bad sizes");
    if (rows == 0) return KTF_OK;                         // (an empty tensor: null pointers)
    KTF_REQUIRE(x && y, "ktf_affine_act_f32: null argument");
    KTF_REQUIRE(act >= KTF_ACT_NONE && act < KTF_ACT_SOFTMAX, "ktf_affine_act_f32: bad activation %d (KTF_ACT_SOFTMAX: ktf_activation_f32)", act);
    const int64_t total = rows * D;
    if (total == 0) return KTF_OK;
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(affine_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, total, D, act, scale, shift, y);
    KTF_CHECK_LAUNCH("ktf_affine_act_f32");
    return KTF_OK;
}

extern "C" int ktf_convert_pad(const void* src, int32_t src_dtype, int64_t rows, int32_t D, int64_t ld_src, void* dst,
                               int32_t dst_dtype, int64_t ld_dst, void* stream) {
    KTF_REQUIRE(rows >= 0 && D > 0 && ld_src >= D && ld_dst >= D, "ktf_convert_pad: bad sizes");
    const int64_t total = rows * ld_dst;
    if (total == 0) return KTF_OK;                       // (an empty tensor: null pointers)
    KTF_REQUIRE(src && dst, "ktf_convert_pad: null argument");
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
#define CP_CASE(SD, ST, DD, DT)                                                                                        \
    if (src_dtype == SD && dst_dtype == DD) {                                                                          \
        hipLaunchKernelGGL((convert_pad_kernel<ST, DT>), dim3(blocks), dim3(256), 0, st, (const ST*)src, rows, D, ld_src, \
                           (DT*)dst, ld_dst);                                                                          \
        launched = true;                                                                                               \
    }
    bool launched = false;
    CP_CASE(KTF_F32, float, KTF_F32, float) CP_CASE(KTF_F32, float, KTF_BF16, unsigned short)
    CP_CASE(KTF_BF16, unsigned short, KTF_F32, float) CP_CASE(KTF_BF16, unsigned short, KTF_BF16, unsigned short)
#undef CP_CASE
    KTF_REQUIRE(launched, "ktf_convert_pad: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
    KTF_CHECK_LAUNCH("ktf_convert_pad");
    return KTF_OK;
}


