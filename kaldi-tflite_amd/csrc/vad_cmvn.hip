// VAD -> per-utterance compaction -> sliding-window CMVN for gfx950.
//
// One 256-thread workgroup owns one utterance (utterances are independent, so a batch is
// B workgroups; no inter-workgroup traffic). The energy VAD needs the utterance mean of C0
// (block reduction), a (2*ctx+1)-tap vote with the reference's edge denominators, and a
// block-wide exclusive scan (wave ballots + LDS) to compact the kept frame numbers.
// CMVN evaluates the windowed sums S[s] = sum_{i<N} x[s+i] for every window start s with a
// chunked sliding update (direct sum for the first window of each 32-start chunk, then
// add-new/subtract-old), which is more accurate than the reference's difference of fp32
// cumulative sums and embarrassingly parallel over (chunk, feature).
//
// Replaces: layers/dsp/vad.py:156-203, models/kaldi/xvector_extractor.py:163-165,
//           layers/normalization/cmvn.py:186-250 of the reference.
#include "common.h"

#define VC_THREADS 256
#define VC_WAVES (VC_THREADS / KTF_WAVE)
#define CMVN_CHUNK 32

__device__ __forceinline__ float block_sum(float v, float* red /* VC_WAVES floats in LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < VC_WAVES; ++w) t += red[w];
    return t;
}

// keep[t] of VAD.call for one utterance; feats rows of stride D.
__device__ __forceinline__ bool vad_keep(const float* __restrict__ f, int64_t T, int D, const KtfVadCfg& c, float thr,
                                         int64_t t) {
    const int ctx = c.frames_context;
    if (ctx == 0) return f[t * D + c.energy_coeff] > thr;
    int cnt = 0;
    for (int k = -ctx; k <= ctx; ++k) {
        const int64_t u = t + k;
        if (u >= 0 && u < T) cnt += (f[u * D + c.energy_coeff] > thr) ? 1 : 0;
    }
    // vad.py:124-135,187-193: denominators at the edges = number of taps inside the sequence
    int den = 2 * ctx + 1;
    if (t < ctx) den = ctx + 1 + (int)t;
    else if (t >= T - ctx) den = ctx + (int)(T - t);
    return ((float)cnt / (float)den) >= c.proportion_threshold;
}

__device__ __forceinline__ float vad_threshold(const float* __restrict__ f, int64_t T, int D, const KtfVadCfg& c,
                                               float* red) {
    float thr = c.energy_threshold;
    if (c.energy_mean_scale > 0.0f) {
        float s = 0.0f;
        for (int64_t t = threadIdx.x; t < T; t += VC_THREADS) s += f[t * D + c.energy_coeff];
        const float mean = block_sum(s, red) / (float)T;
        thr += c.energy_mean_scale * mean;
    }
    return thr;
}

// Compacts kept frame numbers of one utterance into idx[0..count); returns count (block-uniform).
__device__ int vad_compact(const float* __restrict__ f, int64_t T, int D, const KtfVadCfg& c, float thr,
                           int32_t* __restrict__ idx, int* scan /* VC_WAVES+1 ints in LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int64_t t0 = 0; t0 < T; t0 += VC_THREADS) {
        const int64_t t = t0 + threadIdx.x;
        const bool keep = (t < T) && vad_keep(f, T, D, c, thr, t);
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) scan[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < VC_WAVES; ++w) {
            const int cw = scan[w];
            if (w < wave) woff += cw;
            tot += cw;
        }
        if (keep) idx[base + woff + before] = (int32_t)t;
        base += tot;
    }
    return base;
}

template <typename OutT>
__device__ __forceinline__ void store_out(OutT* p, float v);
template <>
__device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_out<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }

// CMVN of one utterance: rows r < len, row r read at x[(idx ? idx[r] : r) * ldx + d].
// work: 2 * max(len - N + 1, 1) * D floats (window sums, and sums of squares when norm_vars).
template <typename OutT>
__device__ void cmvn_block(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ idx, int len, int D,
                           const KtfCmvnCfg& c, OutT* __restrict__ out, int64_t ldo, float* __restrict__ work,
                           float* red, int* out_len) {
    const int N = c.window;
    const int tid = threadIdx.x;
    const int ldo_i = (int)ldo;
    if (len <= N) {
        // cmvn.py:214-222: statistics over all frames
        for (int d0 = 0; d0 < D; d0 += 1) {
            float s = 0.0f, s2 = 0.0f;
            for (int r = tid; r < len; r += VC_THREADS) {
                const float v = x[(int64_t)(idx ? idx[r] : r) * ldx + d0];
                s += v;
                s2 += v * v;
            }
            const float sum = block_sum(s, red);
            const float sum2 = c.norm_vars ? block_sum(s2, red) : 0.0f;
            if (tid == 0) {
                work[d0] = sum / (float)len;
                work[D + d0] = c.norm_vars ? sqrtf(sum2 / (float)len - (sum / (float)len) * (sum / (float)len)) : 1.0f;
            }
        }
        __syncthreads();
        for (int e = tid; e < len * ldo_i; e += VC_THREADS) {
            const int r = e / ldo_i, d = e - r * ldo_i;
            float v = 0.0f;
            if (d < D) {
                v = x[(int64_t)(idx ? idx[r] : r) * ldx + d] - work[d];
                if (c.norm_vars) v = v / work[D + d];
            }
            store_out<OutT>(out + (int64_t)r * ldo + d, v);
        }
        if (out_len && tid == 0) *out_len = len;
        return;
    }
    // window sums for every start s in [0, len-N]
    const int nstart = len - N + 1;
    const int nchunk = (nstart + CMVN_CHUNK - 1) / CMVN_CHUNK;
    float* S = work;
    float* S2 = work + (int64_t)nstart * D;
    for (int item = tid; item < nchunk * D; item += VC_THREADS) {
        const int ch = item / D, d = item - ch * D;
        const int s0 = ch * CMVN_CHUNK;
        const int s1 = min(s0 + CMVN_CHUNK, nstart);
        float a = 0.0f, a2 = 0.0f;
        for (int i = 0; i < N; ++i) {
            const float v = x[(int64_t)(idx ? idx[s0 + i] : s0 + i) * ldx + d];
            a += v;
            a2 += v * v;
        }
        S[(int64_t)s0 * D + d] = a;
        if (c.norm_vars) S2[(int64_t)s0 * D + d] = a2;
        for (int s = s0 + 1; s < s1; ++s) {
            const float vn = x[(int64_t)(idx ? idx[s + N - 1] : s + N - 1) * ldx + d];
            const float vo = x[(int64_t)(idx ? idx[s - 1] : s - 1) * ldx + d];
            a += vn - vo;
            a2 += vn * vn - vo * vo;
            S[(int64_t)s * D + d] = a;
            if (c.norm_vars) S2[(int64_t)s * D + d] = a2;
        }
    }
    __syncthreads();
    // cmvn.py:172-182: frame t uses the window starting at clamp(t - N/2, 0, len - N); VALID keeps [N/2, len-(N-1)/2)
    const int a = c.valid ? N / 2 : 0;
    const int b = c.valid ? len - (N - 1) / 2 : len;
    const int nout = b - a;
    for (int e = tid; e < nout * ldo_i; e += VC_THREADS) {
        const int j = e / ldo_i, d = e - j * ldo_i;
        const int t = a + j;
        float v = 0.0f;
        if (d < D) {
            int s = t - N / 2;
            s = s < 0 ? 0 : (s > len - N ? len - N : s);
            const float mean = S[(int64_t)s * D + d] / (float)N;
            v = x[(int64_t)(idx ? idx[t] : t) * ldx + d] - mean;
            if (c.norm_vars) {
                const float std = sqrtf(S2[(int64_t)s * D + d] / (float)N - mean * mean);
                v = v / std;
            }
        }
        store_out<OutT>(out + (int64_t)j * ldo + d, v);
    }
    if (out_len && tid == 0) *out_len = nout;
}

__global__ __launch_bounds__(VC_THREADS) void vad_mask_kernel(const float* __restrict__ feats, int64_t T, int D,
                                                              KtfVadCfg c, float* __restrict__ mask) {
    __shared__ float red[VC_WAVES];
    const float* f = feats + (int64_t)blockIdx.x * T * D;
    const float thr = vad_threshold(f, T, D, c, red);
    for (int64_t t = threadIdx.x; t < T; t += VC_THREADS)
        mask[(int64_t)blockIdx.x * T + t] = vad_keep(f, T, D, c, thr, t) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(VC_THREADS) void vad_index_kernel(const float* __restrict__ feats, int64_t T, int D,
                                                               KtfVadCfg c, int32_t* __restrict__ idx,
                                                               int32_t* __restrict__ lens) {
    __shared__ float red[VC_WAVES];
    __shared__ int scan[VC_WAVES + 1];
    const float* f = feats + (int64_t)blockIdx.x * T * D;
    const float thr = vad_threshold(f, T, D, c, red);
    const int n = vad_compact(f, T, D, c, thr, idx + (int64_t)blockIdx.x * T, scan);
    if (threadIdx.x == 0) lens[blockIdx.x] = n;
}

__global__ __launch_bounds__(VC_THREADS) void cmvn_kernel(const float* __restrict__ x, int64_t T, int D, int64_t ldx,
                                                          const int32_t* __restrict__ lens, KtfCmvnCfg c,
                                                          float* __restrict__ out, int64_t ldo,
                                                          int32_t* __restrict__ out_lens, float* __restrict__ work) {
    __shared__ float red[VC_WAVES];
    const int b = blockIdx.x;
    const int len = lens ? lens[b] : (int)T;
    int* ol = out_lens ? out_lens + b : nullptr;
    cmvn_block<float>(x + (int64_t)b * T * ldx, ldx, nullptr, len, D, c, out + (int64_t)b * T * ldo, ldo,
                      work + (int64_t)b * T * 2 * D, red, ol);
}

template <typename OutT>
__global__ __launch_bounds__(VC_THREADS) void vad_cmvn_kernel(const float* __restrict__ feats, int64_t T, int D,
                                                              KtfVadCfg vc, KtfCmvnCfg cc, OutT* __restrict__ out,
                                                              int64_t ldo, int32_t* __restrict__ lens,
                                                              int32_t* __restrict__ idx_work,
                                                              float* __restrict__ work) {
    __shared__ float red[VC_WAVES];
    __shared__ int scan[VC_WAVES + 1];
    const int b = blockIdx.x;
    const float* f = feats + (int64_t)b * T * D;
    int32_t* idx = idx_work + (int64_t)b * T;
    const float thr = vad_threshold(f, T, D, vc, red);
    const int n = vad_compact(f, T, D, vc, thr, idx, scan);
    __syncthreads();  // idx[] written by this workgroup is read below by other threads of it
    int* ol = lens + b;
    cmvn_block<OutT>(f, D, idx, n, D, cc, out + (int64_t)b * T * ldo, ldo, work + (int64_t)b * T * 2 * D, red, ol);
}

static int check_vad(const char* who, const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* c) {
    KTF_REQUIRE(feats && c, "%s: null argument", who);
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0, "%s: bad sizes", who);
    KTF_REQUIRE(c->energy_coeff >= 0 && c->energy_coeff < D, "%s: energy_coeff %d outside [0,%d)", who, c->energy_coeff, D);
    KTF_REQUIRE(c->frames_context >= 0, "%s: frames_context must be >= 0", who);
    KTF_REQUIRE(c->energy_mean_scale >= 0.0f, "%s: energy_mean_scale must be >= 0", who);
    KTF_REQUIRE(T == 0 || T >= 2 * (int64_t)c->frames_context, "%s: T=%lld shorter than 2*frames_context", who, (long long)T);
    KTF_REQUIRE(T < (1ll << 31), "%s: T too large", who);
    return KTF_OK;
}

extern "C" int ktf_vad_mask_f32(const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* cfg,
                                float* mask, void* stream) {
    int rc = check_vad("ktf_vad_mask_f32", feats, B, T, D, cfg);
    if (rc) return rc;
    KTF_REQUIRE(mask, "ktf_vad_mask_f32: null mask");
    if (B * T == 0) return KTF_OK;
    hipLaunchKernelGGL(vad_mask_kernel, dim3((unsigned)B), dim3(VC_THREADS), 0, (hipStream_t)stream, feats, T, D, *cfg, mask);
    KTF_CHECK_LAUNCH("ktf_vad_mask_f32");
    return KTF_OK;
}

extern "C" int ktf_vad_index(const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* cfg, int32_t* idx,
                             int32_t* lens, void* stream) {
    int rc = check_vad("ktf_vad_index", feats, B, T, D, cfg);
    if (rc) return rc;
    KTF_REQUIRE(idx && lens, "ktf_vad_index: null output");
    if (B == 0) return KTF_OK;
    hipLaunchKernelGGL(vad_index_kernel, dim3((unsigned)B), dim3(VC_THREADS), 0, (hipStream_t)stream, feats, T, D, *cfg, idx, lens);
    KTF_CHECK_LAUNCH("ktf_vad_index");
    return KTF_OK;
}

static int check_cmvn(const char* who, const KtfCmvnCfg* c) {
    KTF_REQUIRE(c, "%s: null cmvn config", who);
    KTF_REQUIRE(c->window > 0, "%s: window must be > 0", who);
    return KTF_OK;
}

extern "C" int ktf_cmvn_f32(const float* x, int64_t B, int64_t T, int32_t D, int64_t ldx, const int32_t* lens,
                            const KtfCmvnCfg* cfg, float* out, int64_t ldo, int32_t* out_lens, float* work,
                            void* stream) {
    int rc = check_cmvn("ktf_cmvn_f32", cfg);
    if (rc) return rc;
    KTF_REQUIRE(x && out && work, "ktf_cmvn_f32: null argument");
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0 && ldx >= D && ldo >= D, "ktf_cmvn_f32: bad sizes");
    KTF_REQUIRE(T < (1ll << 31) / (ldo > 0 ? ldo : 1), "ktf_cmvn_f32: T*ldo too large");
    if (B * T == 0) return KTF_OK;
    hipLaunchKernelGGL(cmvn_kernel, dim3((unsigned)B), dim3(VC_THREADS), 0, (hipStream_t)stream, x, T, D, ldx, lens, *cfg,
                       out, ldo, out_lens, work);
    KTF_CHECK_LAUNCH("ktf_cmvn_f32");
    return KTF_OK;
}

extern "C" int ktf_vad_cmvn(const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* vad,
                            const KtfCmvnCfg* cmvn, void* out, int32_t out_dtype, int64_t ldo, int32_t* lens,
                            int32_t* idx_work, float* work, void* stream) {
    int rc = check_vad("ktf_vad_cmvn", feats, B, T, D, vad);
    if (rc) return rc;
    rc = check_cmvn("ktf_vad_cmvn", cmvn);
    if (rc) return rc;
    KTF_REQUIRE(out && lens && idx_work && work, "ktf_vad_cmvn: null argument");
    KTF_REQUIRE(ldo >= D, "ktf_vad_cmvn: ldo < D");
    KTF_REQUIRE(out_dtype == KTF_F32 || out_dtype == KTF_BF16, "ktf_vad_cmvn: bad out_dtype");
    KTF_REQUIRE(T < (1ll << 31) / (ldo > 0 ? ldo : 1), "ktf_vad_cmvn: T*ldo too large");
    if (B == 0) return KTF_OK;
    if (T == 0) {
        (void)hipMemsetAsync(lens, 0, sizeof(int32_t) * B, (hipStream_t)stream);
        return KTF_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == KTF_F32)
        hipLaunchKernelGGL(vad_cmvn_kernel<float>, dim3((unsigned)B), dim3(VC_THREADS), 0, st, feats, T, D, *vad, *cmvn,
                           (float*)out, ldo, lens, idx_work, work);
    else
        hipLaunchKernelGGL(vad_cmvn_kernel<unsigned short>, dim3((unsigned)B), dim3(VC_THREADS), 0, st, feats, T, D, *vad,
                           *cmvn, (unsigned short*)out, ldo, lens, idx_work, work);
    KTF_CHECK_LAUNCH("ktf_vad_cmvn");
    return KTF_OK;
}
