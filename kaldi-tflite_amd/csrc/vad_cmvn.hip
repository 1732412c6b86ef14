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
#include <type_traits>

#define VC_THREADS 1024
#define VC_RG (VC_THREADS / 32)      // row groups of the (row group, 32 columns) thread map
#define VC_GM (2 * VC_RG * 32)        // floats of LDS scratch in front of the staging area
#define VC_WAVES (VC_THREADS / KTF_WAVE)
#define CMVN_CHUNK 32

// phase stamps of workgroup 0 (probe builds only: tools/vc_phase_probe.py)
#define VC_PROBE(k)

// m * m rounded to fp32 before anything is subtracted from it (cmvn.py:206, 222: tf.pow(mean, 2) is a tensor of its own). HIP contracts
// a * b - c into a fused multiply-add by default (and __fmul_rn is a plain product there); with ONE frame the reference's variance is
// exactly 0 and its output 0 / 0, a fused form leaves the rounding residual of the square instead.
__device__ __forceinline__ float sq_rounded(float m) {
#pragma clang fp contract(off)
    return m * m;
}

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

// keep[t] of VAD.call for one utterance; e[u * es] = the energy coefficient of frame u (the feature rows themselves: e = feats +
// energy_coeff, es = D; or a copy of that column: es = 1).
__device__ __forceinline__ bool vad_keep(const float* __restrict__ e, int64_t es, int64_t T, const KtfVadCfg& c, float thr,
                                         int64_t t) {
    const int ctx = c.frames_context;
    if (ctx == 0) return e[t * es] > thr;
    int cnt = 0;
    for (int k = -ctx; k <= ctx; ++k) {
        const int64_t u = t + k;
        if (u >= 0 && u < T) cnt += (e[u * es] > thr) ? 1 : 0;
    }
    // vad.py:124-135,187-193: denominators at the edges = number of taps inside the sequence
    int den = 2 * ctx + 1;
    if (T >= 2 * (int64_t)ctx) {
        if (t < ctx) den = ctx + 1 + (int)t;
        else if (t >= T - ctx) den = ctx + (int)(T - t);
    } else {
        // fewer than 2*ctx frames: the reference scatters the edge sizes ctx+1 .. 2ctx (left) then 2ctx .. ctx+1 (right) at
        // indexes floormod(i + T, T), i = 0 .. ctx-1, -ctx .. -1, in that order, and the LAST write to a frame stands
        bool hit = false;
        for (int j = ctx - 1; j >= 0 && !hit; --j) {
            int64_t i = (T - ctx + j) % T;
            if (i < 0) i += T;
            if (i == t) { den = 2 * ctx - j; hit = true; }
        }
        for (int j = ctx - 1; j >= 0 && !hit; --j)
            if ((int64_t)j % T == t) { den = ctx + 1 + j; hit = true; }
    }
    return ((float)cnt / (float)den) >= c.proportion_threshold;
}

// (col: when given, the energy column is also copied there -- T floats of LDS the vote then reads instead of the feature rows)
__device__ __forceinline__ float vad_threshold(const float* __restrict__ f, int64_t T, int D, const KtfVadCfg& c,
                                               float* red, float* __restrict__ col = nullptr) {
    float thr = c.energy_threshold;
    if (c.energy_mean_scale > 0.0f || col) {
        float s = 0.0f;
        for (int64_t t = threadIdx.x; t < T; t += VC_THREADS) {
            const float v = f[t * D + c.energy_coeff];
            if (col) col[t] = v;
            s += v;
        }
        if (c.energy_mean_scale > 0.0f) {
            const float mean = block_sum(s, red) / (float)T;
            thr += c.energy_mean_scale * mean;
        } else {
            __syncthreads();
        }
    }
    return thr;
}

// Compacts kept frame numbers of one utterance into idx[0..count) (and into idx2, if given); returns count (block-uniform).
__device__ int vad_compact(const float* __restrict__ e, int64_t es, int64_t T, const KtfVadCfg& c, float thr,
                           int32_t* __restrict__ idx, int32_t* __restrict__ idx2, int* scan /* VC_WAVES+1 ints in LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int64_t t0 = 0; t0 < T; t0 += VC_THREADS) {
        const int64_t t = t0 + threadIdx.x;
        const bool keep = (t < T) && vad_keep(e, es, T, c, thr, t);
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
        if (keep) {
            idx[base + woff + before] = (int32_t)t;
            if (idx2) idx2[base + woff + before] = (int32_t)t;
        }
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

// CMVN of one utterance: rows r < len, row r read at x[(inv ? inv[r] : r) * ldx + d].
// The (compacted) rows are first staged contiguously into `xs` (len*D floats: LDS when the utterance fits, else the
// caller's global workspace), which removes the idx indirection and the global-memory latency from the sliding loops.
// Then every (chunk of CMVN_CHUNK window starts, column) item computes its first window sum directly, slides it, and
// writes the normalised frames itself — no window-sum array, no second pass. Pad columns [D, ldo) are written as zeros.
// gm: VC_GM floats of LDS scratch for the whole-utterance branch.
// split / nsplit: the workgroup is one of nsplit that share the utterance (small batches: one workgroup per utterance leaves 255 CUs
// idle and its window phase is bound by ONE CU's vector issue). A split owns a contiguous range of the window-start chunks; it stages
// only the rows its windows read and writes only its frames -- every value is computed exactly as the unsplit workgroup computes it.
template <typename OutT>
__device__ __forceinline__ void cmvn_block(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ inv, int len, int D,
                           const KtfCmvnCfg& c, OutT* __restrict__ out, int64_t ldo, float* __restrict__ xs, float* gm,
                           int* out_len, float* __restrict__ bs = nullptr, int split = 0, int nsplit = 1) {
    const int N = c.window;
    const int tid = threadIdx.x;
    const int ldo_i = (int)ldo;
    if (len <= N && split) return;                         // whole-utterance statistics: the first split does all of it
    // this split's chunks [cA, cB) of window starts and the rows [r_lo, r_hi) they read
    int cA = 0, cB = 0, r_lo = 0, r_hi = len;
    if (len > N) {
        const int nstart_ = len - N + 1, nchunk_ = (nstart_ + CMVN_CHUNK - 1) / CMVN_CHUNK;
        cA = (int)((int64_t)nchunk_ * split / nsplit);
        cB = (int)((int64_t)nchunk_ * (split + 1) / nsplit);
        if (cB <= cA) return;
        r_lo = cA * CMVN_CHUNK;
        r_hi = min(cB * CMVN_CHUNK, nstart_) + N - 1;
    }
    // staging: the rows this split reads, up to 32 independent global loads in flight per thread (one load behind each store
    // would serialise the ~1 us latencies); kept frames are mostly consecutive, so a row group's loads stay coalesced
    {
        const int rs = tid >> 5, dl = tid & 31;
        for (int d0 = 0; d0 < D; d0 += 32) {
            const int d = d0 + dl;
            if (d < D)
                for (int r = r_lo + rs; r < r_hi; r += VC_RG * 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int rr = r + u * VC_RG;
                        const bool ok = rr < r_hi;
                        const int t = ok ? (inv ? inv[rr] : rr) : 0;
                        v[u] = ok ? x[(int64_t)t * ldx + d] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int rr = r + u * VC_RG;
                        if (rr < r_hi) xs[rr * D + d] = v[u];
                    }
                }
        }
    }
    __syncthreads();
    VC_PROBE(3)
    if (len <= N) {
        // cmvn.py:214-222: statistics over all frames. VC_RG row groups x 32 columns per pass.
        const int rg = tid >> 5, dl = tid & 31;
        for (int d0 = 0; d0 < ldo_i; d0 += 32) {
            const int d = d0 + dl;
            float s = 0.0f, s2 = 0.0f;
            if (d < D)
                for (int r = rg; r < len; r += VC_RG) {
                    const float v = xs[r * D + d];
                    s += v;
                    s2 += v * v;
                }
            gm[rg * 32 + dl] = s;
            gm[VC_RG * 32 + rg * 32 + dl] = s2;
            __syncthreads();
            float mean = 0.0f, sd = 1.0f;
            {
                float ts = 0.0f, ts2 = 0.0f;
#pragma unroll
                for (int g = 0; g < VC_RG; ++g) {
                    ts += gm[g * 32 + dl];
                    ts2 += gm[VC_RG * 32 + g * 32 + dl];
                }
                mean = ts / (float)len;
                if (c.norm_vars) sd = sqrtf(ts2 / (float)len - sq_rounded(mean));      // (cmvn.py:206, 222: the square is rounded before the
                                                                                             // subtraction, no fused multiply-add: one frame -> exactly 0)
            }
            // VALID keeps the frames [N/2, len - (N-1)/2) (cmvn.py:238-243): none of an utterance shorter than the window, one of an
            // utterance exactly as long
            const int r0 = c.valid ? N / 2 : 0, nout = c.valid ? (len == N ? 1 : 0) : len;
            for (int r = rg; r < nout; r += VC_RG) {
                if (d < ldo_i) {
                    float v = 0.0f;
                    if (d < D) {
                        v = xs[(r0 + r) * D + d] - mean;
                        if (c.norm_vars) v = v / sd;
                    }
                    store_out<OutT>(out + (int64_t)r * ldo + d, v);
                }
            }
            __syncthreads();
        }
        if (out_len && tid == 0) *out_len = c.valid ? (len == N ? 1 : 0) : len;
        return;
    }
    // cmvn.py:172-182: frame t uses the window starting at clamp(t - N/2, 0, len - N); VALID keeps [N/2, len-(N-1)/2)
    const int nstart = len - N + 1;
    const int nchunk = (nstart + CMVN_CHUNK - 1) / CMVN_CHUNK;
    const int half = N / 2;
    const float fN = (float)N;
    // sums of the CMVN_CHUNK-row blocks of every column (bs: [2][nblk][ldo] floats of LDS, when the launcher found room):
    // the first window of a chunk starts on a block boundary, so its sum is N/CMVN_CHUNK block sums plus a short tail
    // instead of a chain of N dependent LDS reads per item
    const int nblk = len / CMVN_CHUNK;                 // complete blocks
    if (bs) {
        float* bs2 = bs + (size_t)((len + CMVN_CHUNK - 1) / CMVN_CHUNK) * ldo_i;
        const int kB = min(nblk, cB - 1 + N / CMVN_CHUNK);         // the blocks this split's first windows are made of
        for (int item = cA * ldo_i + tid; item < kB * ldo_i; item += VC_THREADS) {
            const int k = item / ldo_i, d = item - k * ldo_i;
            float a = 0.0f, a2 = 0.0f;
            if (d < D) {
                const float* p = xs + k * CMVN_CHUNK * D + d;
#pragma unroll 8
                for (int i = 0; i < CMVN_CHUNK; ++i) {
                    const float v = p[i * D];
                    a += v;
                    a2 += v * v;
                }
            }
            bs[item] = a;
            bs2[item] = a2;
        }
        __syncthreads();
    }
    VC_PROBE(4)
    (void)nchunk;
    for (int item = cA * ldo_i + tid; item < cB * ldo_i; item += VC_THREADS) {
        const int ch = item / ldo_i, d = item - ch * ldo_i;
        const int s0 = ch * CMVN_CHUNK;
        const int s1 = min(s0 + CMVN_CHUNK, nstart);
        const bool real = d < D;
        float a = 0.0f, a2 = 0.0f;
        if (real) {
            int i = 0;
            if (bs) {
                const float* bs2 = bs + (size_t)((len + CMVN_CHUNK - 1) / CMVN_CHUNK) * ldo_i;
                const int nb = N / CMVN_CHUNK;       // whole blocks inside the window (all complete: s0 + N <= len)
#pragma unroll 4
                for (int k = 0; k < nb; ++k) {
                    a += bs[(ch + k) * ldo_i + d];
                    a2 += bs2[(ch + k) * ldo_i + d];
                }
                i = nb * CMVN_CHUNK;
            }
            const float* p = xs + s0 * D + d;
#pragma unroll 4
            for (; i < N; ++i) {
                const float v = p[i * D];
                a += v;
                a2 += v * v;
            }
        }
        // the slide, fully unrolled and branch-free but for the store: eight starts' operands are read ahead of their arithmetic,
        // ~28 instructions per start. (A loop over the starts with its conditions inside compiles to ~100 per start, most of them
        // branches, and with one or two waves of items per SIMD the instruction count IS the time: 7.5 of the kernel's 20 us.)
        // Pad columns run the arithmetic on column 0 and store zeros; an utterance's last chunk may be short: the starts behind
        // its end repeat the last one's operands and store nothing.
        {
            const int cnt = s1 - s0;
            const int dd = real ? d : 0;
            const float* pn = xs + (s0 + N - 1) * D + dd;          // frame entering the window of start s0 + i: pn[i * D]
            const float* po = xs + (s0 - 1) * D + dd;              // frame leaving it: po[i * D] (i >= 1)
            const float* pc = xs + (s0 + half) * D + dd;           // the frame the window is centred on
            OutT* op = out + (s0 + (c.valid ? 0 : half)) * ldo_i + d;      // (T * ldo < 2^31: the launcher checks)
            const int last = (cnt - 1) * D;
            auto chunk = [&](auto nv_tag) {
                constexpr bool NV = decltype(nv_tag)::value;
                float mean = 0.0f, sd = 1.0f, mean0 = 0.0f, sd0 = 1.0f, meanl = 0.0f, sdl = 1.0f;
#pragma unroll
                for (int q = 0; q < CMVN_CHUNK / 8; ++q) {
                    float vn[8], vo[8], xc[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int i = q * 8 + k;
                        const int o = min(i * D, last);
                        vn[k] = i ? pn[o] : 0.0f;
                        vo[k] = i ? po[o] : 0.0f;
                        xc[k] = pc[o];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int i = q * 8 + k;
                        if (i) {
                            a += vn[k] - vo[k];
                            if (NV) a2 += vn[k] * vn[k] - vo[k] * vo[k];
                        }
                        mean = a / fN;
                        if (NV) sd = sqrtf(a2 / fN - sq_rounded(mean));
                        if (i == 0) { mean0 = mean; sd0 = sd; }
                        if (i == cnt - 1) { meanl = mean; sdl = sd; }
                        float v = xc[k] - mean;
                        if (NV) v = v / sd;
                        if (i < cnt) store_out<OutT>(op + i * ldo_i, real ? v : 0.0f);
                    }
                }
                // the first / last window also serve the N/2 edge frames before / after it (SAME): their statistics are parked
                // in LDS and those ~N frames are written by the whole workgroup below
                if (!c.valid) {
                    if (s0 == 0) { gm[d] = real ? mean0 : 0.0f; gm[ldo_i + d] = real ? sd0 : 1.0f; }
                    if (s1 == nstart) { gm[2 * ldo_i + d] = real ? meanl : 0.0f; gm[3 * ldo_i + d] = real ? sdl : 1.0f; }
                }
            };
            if (c.norm_vars) chunk(std::true_type{});
            else chunk(std::false_type{});
        }
    }
    VC_PROBE(5)
    if (!c.valid) {
        __syncthreads();
        const int n_head = half;                              // frames [0, half) use the first window
        const int t_tail = nstart + half;                     // frames [t_tail, len) use the last window
        const int e_lo = cA == 0 ? 0 : n_head;                // (the split that owns the first / last chunk holds its statistics)
        const int e_hi = cB == nchunk ? n_head + (len - t_tail) : n_head;
        for (int e = e_lo * ldo_i + tid; e < e_hi * ldo_i; e += VC_THREADS) {
            const int k = e / ldo_i, d = e - k * ldo_i;
            const bool tail = k >= n_head;
            const int t = tail ? t_tail + (k - n_head) : k;
            float v = 0.0f;
            if (d < D) {
                v = xs[t * D + d] - gm[(tail ? 2 * ldo_i : 0) + d];
                if (c.norm_vars) v = v / gm[(tail ? 3 * ldo_i : ldo_i) + d];
            }
            store_out<OutT>(out + (int64_t)t * ldo + d, v);
        }
    }
    VC_PROBE(6)
    if (out_len && tid == 0 && cA == 0) *out_len = c.valid ? nstart : len;     // (the split that owns the first chunk)
}

__global__ __launch_bounds__(VC_THREADS) void vad_mask_kernel(const float* __restrict__ feats, int64_t T, int D,
                                                              KtfVadCfg c, float* __restrict__ mask) {
    __shared__ float red[VC_WAVES];
    const float* f = feats + (int64_t)blockIdx.x * T * D;
    const float thr = vad_threshold(f, T, D, c, red);
    for (int64_t t = threadIdx.x; t < T; t += VC_THREADS)
        mask[(int64_t)blockIdx.x * T + t] = vad_keep(f + c.energy_coeff, D, T, c, thr, t) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(VC_THREADS) void vad_index_kernel(const float* __restrict__ feats, int64_t T, int D,
                                                               KtfVadCfg c, int32_t* __restrict__ idx,
                                                               int32_t* __restrict__ lens) {
    __shared__ float red[VC_WAVES];
    __shared__ int scan[VC_WAVES + 1];
    const float* f = feats + (int64_t)blockIdx.x * T * D;
    const float thr = vad_threshold(f, T, D, c, red);
    const int n = vad_compact(f + c.energy_coeff, D, T, c, thr, idx + (int64_t)blockIdx.x * T, nullptr, scan);
    if (threadIdx.x == 0) lens[blockIdx.x] = n;
}

// LDSF: the utterance is staged in LDS -- a pointer that is LDS in one launch and the global workspace in another compiles to flat
// loads (80 of them in the window phase, ~300 ns per dependent access); the two forms are two instantiations instead.
template <bool LDSF>
__global__ __launch_bounds__(VC_THREADS) void cmvn_kernel(const float* __restrict__ x, int64_t T, int D, int64_t ldx,
                                                          const int32_t* __restrict__ lens, KtfCmvnCfg c,
                                                          float* __restrict__ out, int64_t ldo,
                                                          int32_t* __restrict__ out_lens, float* __restrict__ work,
                                                          int64_t stage_floats, int64_t bs_floats) {
    extern __shared__ __attribute__((aligned(16))) float vc_lds[];
    float* gm = vc_lds;                      // VC_GM floats
    float* bsp = bs_floats ? vc_lds + VC_GM : nullptr;
    float* stage = vc_lds + VC_GM + bs_floats;
    const int b = blockIdx.x;
    const int len = lens ? lens[b] : (int)T;
    int* ol = out_lens ? out_lens + b : nullptr;
    float* xs = LDSF ? stage : work + (int64_t)b * T * 2 * D;
    cmvn_block<float>(x + (int64_t)b * T * ldx, ldx, nullptr, len, D, c, out + (int64_t)b * T * ldo, ldo, xs, gm, ol, bsp);
}

template <typename OutT, bool LDSF>
__global__ __launch_bounds__(VC_THREADS) void vad_cmvn_kernel(const float* __restrict__ feats, int64_t T, int D,
                                                              KtfVadCfg vc, KtfCmvnCfg cc, OutT* __restrict__ out,
                                                              int64_t ldo, int32_t* __restrict__ lens,
                                                              int32_t* __restrict__ idx_work,
                                                              float* __restrict__ work, int64_t stage_floats,
                                                              int64_t bs_floats, int64_t pos_ints, int64_t col_floats) {
    extern __shared__ __attribute__((aligned(16))) float vc_lds[];
    float* gm = vc_lds;                      // VC_GM floats (also the reduction scratch of the VAD phase)
    const int b = blockIdx.x;
    const int split = blockIdx.y, nsplit = gridDim.y;     // (every split repeats the VAD: it needs the whole frame -> row map)
    int32_t* idx = idx_work + (int64_t)b * T;
    // compacted row -> frame: T ints of LDS (and, from the first split, a copy in the caller's idx_work); a recording too long for
    // that (pos_ints == 0, > 38,400 frames) keeps only the copy in idx_work (written and read by this workgroup: same CU, same L1)
    int32_t* inv = (LDSF || pos_ints) ? reinterpret_cast<int32_t*>(vc_lds + VC_GM) : idx;
    float* bsp = bs_floats ? vc_lds + VC_GM + pos_ints : nullptr;
    float* col = col_floats ? vc_lds + VC_GM + pos_ints + bs_floats : nullptr;       // the energy column (vote input)
    float* stage = vc_lds + VC_GM + pos_ints + bs_floats + col_floats;
    float* red = gm;
    int* scan = reinterpret_cast<int*>(gm + 64);
    const float* f = feats + (int64_t)b * T * D;
    VC_PROBE(0)
    const float thr = vad_threshold(f, T, D, vc, red, col);
    VC_PROBE(1)
    const int n = LDSF ? vad_compact(col, 1, T, vc, thr, inv, split == 0 ? idx : nullptr, scan)
                       : vad_compact(col ? col : f + vc.energy_coeff, col ? 1 : D, T, vc, thr, inv, (pos_ints && split == 0) ? idx : nullptr, scan);
    __syncthreads();  // inv[] written by this workgroup is read below by other threads of it
    VC_PROBE(2)
    int* ol = lens + b;
    float* xs = LDSF ? stage : (((int64_t)n * D <= stage_floats) ? stage : work + (int64_t)b * T * 2 * D);
    cmvn_block<OutT>(f, D, inv, n, D, cc, out + (int64_t)b * T * ldo, ldo, xs, gm, ol, bsp, split, nsplit);
    VC_PROBE(7)
}

static int check_vad(const char* who, const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* c) {
    KTF_REQUIRE(feats && c, "%s: null argument", who);
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0, "%s: bad sizes", who);
    KTF_REQUIRE(c->energy_coeff >= 0 && c->energy_coeff < D, "%s: energy_coeff %d outside [0,%d)", who, c->energy_coeff, D);
    KTF_REQUIRE(c->frames_context >= 0, "%s: frames_context must be >= 0", who);
    KTF_REQUIRE(c->energy_mean_scale >= 0.0f, "%s: energy_mean_scale must be >= 0", who);
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

// floats of LDS used to stage one utterance (0 = stage in the global workspace): whole utterances up to 150 KiB
static int64_t vc_stage_floats(int64_t T, int32_t D) {
    const int64_t need = T * D;
    return (need * 4 <= 148 * 1024) ? need : 0;
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
    KTF_REQUIRE(ldo <= VC_GM / 4, "ktf_cmvn_f32: ldo > %d", VC_GM / 4);
    KTF_REQUIRE(T < (1ll << 31) / (ldo > 0 ? ldo : 1), "ktf_cmvn_f32: T*ldo too large");
    if (B * T == 0) return KTF_OK;
    int64_t stage_floats = vc_stage_floats(T, D);
    int64_t bs_floats = 2 * ((T + CMVN_CHUNK - 1) / CMVN_CHUNK) * ldo;
    if ((VC_GM + bs_floats + stage_floats) * 4 > 158 * 1024) bs_floats = 0;     // block sums only when they fit beside the staged utterance
    const size_t lds = (VC_GM + (size_t)bs_floats + (size_t)stage_floats) * sizeof(float);
    if (stage_floats) {
        KTF_LDS_ONCE(160 * 1024, cmvn_kernel<true>);
        hipLaunchKernelGGL(cmvn_kernel<true>, dim3((unsigned)B), dim3(VC_THREADS), lds, (hipStream_t)stream, x, T, D, ldx, lens, *cfg,
                           out, ldo, out_lens, work, stage_floats, bs_floats);
    } else {
        KTF_LDS_ONCE(160 * 1024, cmvn_kernel<false>);
        hipLaunchKernelGGL(cmvn_kernel<false>, dim3((unsigned)B), dim3(VC_THREADS), lds, (hipStream_t)stream, x, T, D, ldx, lens, *cfg,
                           out, ldo, out_lens, work, stage_floats, bs_floats);
    }
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
    KTF_REQUIRE(ldo >= D && ldo <= VC_GM / 4, "ktf_vad_cmvn: ldo must be in [D, %d]", VC_GM / 4);
    KTF_REQUIRE(out_dtype == KTF_F32 || out_dtype == KTF_BF16, "ktf_vad_cmvn: bad out_dtype");
    KTF_REQUIRE(T < (1ll << 31) / (ldo > 0 ? ldo : 1), "ktf_vad_cmvn: T*ldo too large");
    if (B == 0) return KTF_OK;
    if (T == 0) {
        (void)hipMemsetAsync(lens, 0, sizeof(int32_t) * B, (hipStream_t)stream);
        return KTF_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    // the frame -> compacted-row map lives in LDS while (T + VC_GM) * 4 B <= 158 KiB (utterances up to ~6.5 min at 10 ms);
    // beyond that it lives in idx_work (which then holds the map, not the kept frame numbers)
    int64_t pos_ints = (T + 3) & ~3ll;
    if ((VC_GM + pos_ints) * 4 > 158 * 1024) pos_ints = 0;
    int64_t stage_floats = vc_stage_floats(T, D);
    if ((VC_GM + pos_ints + stage_floats) * 4 > 158 * 1024) stage_floats = 0;
    int64_t bs_floats = 2 * ((T + CMVN_CHUNK - 1) / CMVN_CHUNK) * ldo;
    if ((VC_GM + pos_ints + bs_floats + stage_floats) * 4 > 158 * 1024) bs_floats = 0;
    int64_t col_floats = (T + 3) & ~3ll;
    if ((VC_GM + pos_ints + bs_floats + col_floats + stage_floats) * 4 > 158 * 1024) col_floats = 0;
    const size_t lds = (VC_GM + (size_t)pos_ints + (size_t)bs_floats + (size_t)col_floats + (size_t)stage_floats) * sizeof(float);
    // small batches: up to eight workgroups share an utterance (cmvn_block), while the utterance and its map are staged in LDS
    // (a split workgroup of the global-workspace form would write rows its siblings write too)
    unsigned nsplit = 1;
    if (pos_ints && stage_floats && col_floats) nsplit = (unsigned)(B >= 256 ? 1 : (256 / B > 8 ? 8 : 256 / B));
    const dim3 grid((unsigned)B, nsplit);
    const bool ldsf = pos_ints && stage_floats && col_floats;       // map, energy column and rows all in LDS
#define VC_LAUNCH(OutT, F)                                                                                             \
    {                                                                                                                  \
        KTF_LDS_ONCE(160 * 1024, (vad_cmvn_kernel<OutT, F>));                                                          \
        hipLaunchKernelGGL((vad_cmvn_kernel<OutT, F>), grid, dim3(VC_THREADS), lds, st, feats, T, D, *vad, *cmvn, (OutT*)out, ldo, \
                           lens, idx_work, work, stage_floats, bs_floats, pos_ints, col_floats);                       \
    }
    if (out_dtype == KTF_F32) {
        if (ldsf) VC_LAUNCH(float, true) else VC_LAUNCH(float, false)
    } else {
        if (ldsf) VC_LAUNCH(unsigned short, true) else VC_LAUNCH(unsigned short, false)
    }
#undef VC_LAUNCH
    KTF_CHECK_LAUNCH("ktf_vad_cmvn");
    return KTF_OK;
}

// ------------------------------------------------------------------------------------ per-utterance routing by voiced length
// XvectorExtractor.route_short_utterances: lens (B) -> lens_main (utterances of at least `min_frames` voiced frames keep their length,
// the others 0) and lens_short (the complement; utterances without a voiced frame stay 0 in both), and -- for the host, which
// decides whether the second pass is enqueued at all -- the number of short utterances, written to PINNED HOST memory followed by
// the call's sequence number (system-scope release): the host polls the sequence number while the GPU works through the first
// pass's launches; no copy, no event, no stream synchronisation. One workgroup.
__global__ __launch_bounds__(1024) void route_short_kernel(const int32_t* __restrict__ lens, int64_t B, int32_t min_frames, int32_t* __restrict__ lens_main,
                                                           int32_t* __restrict__ lens_short, int32_t* host_flag, int32_t seq) {
    __shared__ int part[16];
    int n = 0;
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
        const int len = lens[b];
        const bool is_short = len > 0 && len < min_frames;
        lens_main[b] = len >= min_frames ? len : 0;
        lens_short[b] = is_short ? len : 0;
        n += is_short ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0 && host_flag) {
        int tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += part[w];
        __hip_atomic_store(host_flag, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_flag + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" int ktf_route_short(const int32_t* lens, int64_t B, int32_t min_frames, int32_t* lens_main, int32_t* lens_short, int32_t* host_flag,
                               int32_t seq, void* stream) {
    KTF_REQUIRE(lens && lens_main && lens_short, "ktf_route_short: null argument");
    KTF_REQUIRE(B >= 0 && min_frames >= 0, "ktf_route_short: bad size");
    if (B == 0 && !host_flag) return KTF_OK;
    hipLaunchKernelGGL(route_short_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lens, B, min_frames, lens_main, lens_short, host_flag, seq);
    KTF_CHECK_LAUNCH("ktf_route_short");
    return KTF_OK;
}
