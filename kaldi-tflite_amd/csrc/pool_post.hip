// Statistics pooling, x-vector post-processing (mean-sub + LDA + length-norm) and PLDA scoring.
//
// Replaces: layers/stats/stats_pooling.py:179-295, models/kaldi/xvector_extractor.py:174-184,
//           layers/plda/plda.py:163-263 of the reference.
#include "common.h"

// tf.nn.relu of the variance (stats_pooling.py:238, 293): a NaN -- the 0 / 0 of a window without a sampled frame -- stays a NaN (fmax
// would return 0 and the standard deviation sqrt(epsilon))
__device__ __forceinline__ double relu_keep_nan(double v) { return v < 0.0 ? 0.0 : v; }

// ------------------------------------------------------------------------------------ stats pooling (reduce)
// One workgroup = one utterance x CW columns; row groups stride the time axis, each thread owns two adjacent columns
// (8-byte fp32 / 4-byte bf16 or half loads).
template <typename T>
__device__ __forceinline__ float2 load2(const T* p);
template <>
__device__ __forceinline__ float2 load2<float>(const float* p) { return *reinterpret_cast<const float2*>(p); }
template <>
__device__ __forceinline__ float2 load2<unsigned short>(const unsigned short* p) {
    const unsigned v = *reinterpret_cast<const unsigned*>(p);
    return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u));
}
template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load1<unsigned short>(const unsigned short* p) { return bf2f(*p); }

// Column statistics of one utterance. Rows are summed in 64 interleaved groups (row j -> group j mod 64, fp64), the
// groups are combined in ONE fixed association ((g + g+16) + g+32) + g+48 for g = 0..15, then over g -- independent of how
// threads map to groups. Two mappings of the 1024 threads share it and therefore give bit-identical results:
//   CW = 128 columns per workgroup, 16 row groups of 64 lanes, four groups per thread (throughput shape);
//   CW = 32 columns per workgroup, 64 row groups of 16 lanes, one group per thread: four times the workgroups and a
//   quarter of the serial chain per thread when only a few utterances are in flight (single-utterance latency).
#define SP_THREADS 1024
template <typename T, bool VEC2, int CW>
__global__ __launch_bounds__(SP_THREADS) void stats_pool_kernel(const T* __restrict__ x, int64_t Tmax, int D, int64_t ldx,
                                                                const int32_t* __restrict__ lens, int period, int include_std,
                                                                float eps, float* __restrict__ out, int64_t ldo) {
    constexpr int LPR = CW / 2;                  // lanes per row group (two columns per thread)
    constexpr int RG = SP_THREADS / LPR;         // row groups per workgroup: 16 or 64
    constexpr int NG = 64 / RG;                  // canonical groups per thread: 4 or 1
    __shared__ double red[RG][2][CW];            // fp64 sums: E[x^2]-mean^2 of a (near-)constant channel must not cancel to noise
    const int b = blockIdx.y;
    const int len = lens ? min(lens[b], (int)Tmax) : (int)Tmax;
    const int lc = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const int c0 = blockIdx.x * CW + lc * 2;
    const T* xb = x + (int64_t)b * Tmax * ldx;
    double s0[NG], s1[NG], q0[NG], q1[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) s0[g] = s1[g] = q0[g] = q1[g] = 0.;
    const int nrows = len <= 0 ? 0 : (len + period - 1) / period;
    if (c0 < D) {
        const bool two = (c0 + 1 < D);
        const int64_t rstep = (int64_t)period * ldx;
#pragma unroll 2
        for (int j0 = rg; j0 < nrows; j0 += 64) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = j0 + g * RG;
                if (j < nrows) {
                    const T* p = xb + (int64_t)j * rstep + c0;
                    float2 v;
                    if (VEC2 && two) v = load2<T>(p);
                    else { v.x = load1<T>(p); v.y = two ? load1<T>(p + 1) : 0.f; }
                    s0[g] += (double)v.x; s1[g] += (double)v.y;
                    q0[g] += (double)v.x * (double)v.x; q1[g] += (double)v.y * (double)v.y;
                }
            }
        }
    }
    if (NG == 4) {
        red[rg][0][lc * 2] = ((s0[0] + s0[1 % NG]) + s0[2 % NG]) + s0[3 % NG];
        red[rg][0][lc * 2 + 1] = ((s1[0] + s1[1 % NG]) + s1[2 % NG]) + s1[3 % NG];
        red[rg][1][lc * 2] = ((q0[0] + q0[1 % NG]) + q0[2 % NG]) + q0[3 % NG];
        red[rg][1][lc * 2 + 1] = ((q1[0] + q1[1 % NG]) + q1[2 % NG]) + q1[3 % NG];
    } else {
        red[rg][0][lc * 2] = s0[0]; red[rg][0][lc * 2 + 1] = s1[0];
        red[rg][1][lc * 2] = q0[0]; red[rg][1][lc * 2 + 1] = q1[0];
    }
    __syncthreads();
    if (threadIdx.x < CW) {
        const int c = blockIdx.x * CW + threadIdx.x;
        if (c < D) {
            double s = 0., q = 0.;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (NG == 4) {
                    s += red[g % RG][0][threadIdx.x];
                    q += red[g % RG][1][threadIdx.x];
                } else {
                    s += ((red[g % RG][0][threadIdx.x] + red[(g + 16) % RG][0][threadIdx.x]) + red[(g + 32) % RG][0][threadIdx.x]) +
                         red[(g + 48) % RG][0][threadIdx.x];
                    q += ((red[g % RG][1][threadIdx.x] + red[(g + 16) % RG][1][threadIdx.x]) + red[(g + 32) % RG][1][threadIdx.x]) +
                         red[(g + 48) % RG][1][threadIdx.x];
                }
            }
            const double n = (double)nrows;
            const double mean = s / n;
            out[(int64_t)b * ldo + c] = (float)mean;
            if (include_std) {
                const double var = q / n - mean * mean;
                out[(int64_t)b * ldo + D + c] = (float)sqrt(relu_keep_nan(var) + (double)eps);
            }
        }
    }
}

// windowed statistics (fp32): one thread per (b, output row j, column c)
__global__ void stats_pool_windowed_kernel(const float* __restrict__ x, int64_t B, int64_t T, int D, int left, int right,
                                           int in_period, int out_period, int start, int64_t Tout, int include_std,
                                           float eps, float* __restrict__ out) {
    const int od = include_std ? 2 * D : D;
    const int64_t total = B * Tout * D;
    const int rc = (right + 1 > T) ? (int)T : right + 1;  // stats_pooling.py:186-191
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % D);
        const int64_t bj = e / D;
        const int64_t j = bj % Tout, b = bj / Tout;
        const int64_t centre = start + j * out_period;
        double s = 0., q = 0., n = 0.;
        for (int o = left; o < rc; o += in_period) {
            const int64_t t = centre + o;
            if (t >= 0 && t < T) {
                const float v = x[(b * T + t) * D + c];
                s += (double)v; q += (double)v * (double)v; n += 1.;
            }
        }
        const double mean = s / n;
        out[(b * Tout + j) * od + c] = (float)mean;
        if (include_std) out[(b * Tout + j) * od + D + c] = (float)sqrt(relu_keep_nan(q / n - mean * mean) + (double)eps);
    }
}

// ------------------------------------------------------------------------------------ x-vector post-processing
// EB embeddings per workgroup: LDS holds their (x - mean); thread = (output column j, slice `part` of the input dimension): the 512-long
// dot products are split into P partial sums (fixed order -> deterministic) so a single embedding is not a serial chain. A workgroup
// streams the whole LDA matrix through its threads (512 x 150 floats = 300 KiB from L2): with one embedding per workgroup a batch of
// 1024 read it 1024 times -- 0.31 GB of L2 reads, the 29 us the launch took -- so batches beyond one round of workgroups put FOUR
// embeddings behind every A element (one 16-byte LDS read of the four inputs, four multiply-adds). The chains, their split and the
// order of every sum are the same for any EB: the same bits.
#define XP_THREADS 1024
template <int EB>
__global__ __launch_bounds__(XP_THREADS) void xvec_post_kernel(const float* __restrict__ x, int64_t B, int in_dim, int out_dim,
                                                               const float* __restrict__ mean, const float* __restrict__ A,
                                                               const float* __restrict__ off, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // EB x in_dim (input-major: [i][EB]) | EB x out_dim | EB x XP_THREADS partials | EB x 16
    float* xc = sm;
    float* yo = sm + EB * in_dim;
    float* part_s = yo + EB * out_dim;
    float* red = part_s + EB * XP_THREADS;
    const int64_t b0 = (int64_t)blockIdx.x * EB;
    const int nb = (int)(B - b0 < EB ? B - b0 : EB);
    const int tid = threadIdx.x;
    for (int q = tid; q < EB * in_dim; q += XP_THREADS) {
        const int e = q / in_dim, i = q - e * in_dim;        // (consecutive threads read consecutive inputs of one embedding)
        xc[i * EB + e] = e < nb ? x[(b0 + e) * in_dim + i] - (mean ? mean[i] : 0.f) : 0.f;
    }
    __syncthreads();
    int J = (out_dim + 63) & ~63;
    if (J > XP_THREADS) J = XP_THREADS;
    const int P = XP_THREADS / J;
    const int jl = tid % J, part = tid / J;
    const int chunk = (in_dim + P - 1) / P;
    const int i_lo = part * chunk, i_hi = (i_lo + chunk < in_dim) ? i_lo + chunk : in_dim;
    float ss[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) ss[e] = 0.f;
    for (int j0 = 0; j0 < out_dim; j0 += J) {
        const int j = j0 + jl;
        float acc[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) acc[e] = 0.f;
        if (part < P && j < out_dim) {
            const float* ap = A + (int64_t)i_lo * out_dim + j;
#pragma unroll 16                                           // loads of 16 rows in flight; every sum stays one ordered chain
            for (int i = i_lo; i < i_hi; ++i, ap += out_dim) {
                const float a = *ap;
#pragma unroll
                for (int e = 0; e < EB; ++e) acc[e] = fmaf(xc[i * EB + e], a, acc[e]);       // (explicit: as `+= x * a` the vectoriser splits one pair into v_pk_mul / v_pk_add)
            }
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) part_s[e * XP_THREADS + tid] = acc[e];
        __syncthreads();
        if (part == 0 && j < out_dim) {
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                float t = 0.f;
                for (int q = 0; q < P; ++q) t += part_s[e * XP_THREADS + q * J + jl];
                t += off ? off[j] : 0.f;
                yo[e * out_dim + j] = t;
                ss[e] = fmaf(t, t, ss[e]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        const float w = wave_sum(ss[e]);
        if ((tid & 63) == 0) red[e * 16 + (tid >> 6)] = w;
    }
    __syncthreads();
    for (int e = 0; e < nb; ++e) {
        float tot = 0.f;
        for (int w = 0; w < XP_THREADS / 64; ++w) tot += red[e * 16 + w];
        const float ratio = sqrtf(tot) / sqrtf((float)out_dim);   // xvector_extractor.py:178-181
        for (int jj = tid; jj < out_dim; jj += XP_THREADS) y[(b0 + e) * out_dim + jj] = yo[e * out_dim + jj] / ratio;
    }
}

// ------------------------------------------------------------------------------------ fused tail (a11 finalize + tdnn6 + a12)
// pooled statistics -> affine after the pooling (tdnn6: `units` x `in_dim`, sequential.py:68-79 "stats -> tdnn6") -> x - mean ->
// LDA (+ offset) -> length normalisation (xvector_extractor.py:174-184), ONE launch instead of finalize + GEMM + post.
// Per utterance XT_NBLK workgroups: each (1) builds the pooled vector in LDS (from fp32 pooled rows, or from the fp64 slot sums
// of the fused pooling: the finalize of stats_pooling.py:231-240, slots added in block order), (2) computes its slice of the
// affine's units -- one wave per unit, lanes over K in 16-byte steps, one fixed butterfly -> deterministic --, (3) writes the
// slice's contribution to the 128 LDA outputs into its own slot; the workgroup that draws the last ticket adds the slots IN SLOT
// ORDER, adds the offset and normalises. Same kernel for one utterance and for a batch, so batch == single bit for bit; a
// single utterance spreads over XT_NBLK CUs instead of running 3000-long fmaf chains on one.
// Hand-off = the agent-scope release / ticket / acquire recipe of cdna_hip_programming.md section 6 (valid for any placement of
// the workgroups over XCDs); the last arriver resets the ticket counter (the host hands in a zeroed array once).
#define XT_NBLK 64
#define XT_THREADS 256
// utterance b: the XT_NBLK slots added IN SLOT ORDER, + offset, length normalisation (xvector_extractor.py:178-181); whole workgroup
__device__ __forceinline__ void xt_reduce(const float* __restrict__ partial, const float* __restrict__ off, float* __restrict__ y, int64_t b,
                                          int out_dim, float* ys, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float ss = 0.0f;
    for (int c = tid; c < out_dim; c += XT_THREADS) {
        float t = 0.0f;
        for (int k = 0; k < XT_NBLK; ++k) t += partial[(b * XT_NBLK + k) * out_dim + c];
        t += off ? off[c] : 0.0f;
        ys[c] = t;
        ss += t * t;
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.0f;
    for (int w = 0; w < XT_THREADS / 64; ++w) tot += red[w];
    const float ratio = sqrtf(tot) / sqrtf((float)out_dim);
    for (int c = tid; c < out_dim; c += XT_THREADS) y[b * out_dim + c] = ys[c] / ratio;
    __syncthreads();
}
#define XT_MAXIT 12                                       // 16-byte K steps per lane: in_dim <= 3072
// Grid (XT_NBLK unit slices, groups of G utterances). A workgroup keeps its slice of W in REGISTERS (two units per wave, 12 x 4
// floats per lane and unit) and walks its G utterances: W is read once per (slice, group), so the same kernel serves one
// utterance (G = 1: 64 workgroups spread the 6 MB of W over 64 CUs) and a large batch (G = 32: W traffic / 32). The arithmetic of
// an utterance does not depend on G or B.
__global__ __launch_bounds__(XT_THREADS) void xvec_tail_kernel(const float* __restrict__ pooled, int64_t ld_pooled, const double* __restrict__ sums,
                                                               int64_t slots, int slot_rows, const int32_t* __restrict__ lens, int64_t T, int D, int include_std,
                                                               float eps, const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
                                                               int in_dim, int units, const float* __restrict__ mean, const float* __restrict__ A,
                                                               const float* __restrict__ off, int out_dim, float* __restrict__ partial,
                                                               unsigned* __restrict__ counters, float* __restrict__ y, float* __restrict__ h_out,
                                                               int64_t B, int G, int two_phase, int skip_empty) {
    extern __shared__ __attribute__((aligned(16))) float xt_sm[];          // in_dim_pad | 16 unit outputs | out_dim | 8
    const int in_pad = (in_dim + 3) & ~3;
    float* xs = xt_sm;
    float* hs = xs + XT_MAXIT * 256;
    float* ys = hs + 16;
    float* red = ys + out_dim;
    const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (skip_empty && lens) {                                   // KTF_TAIL_SKIP_EMPTY: a group without a single voiced utterance leaves at once
        const int64_t lo = (int64_t)blockIdx.y * G, hi = lo + G < B ? lo + G : B;
        bool any = false;
        for (int64_t b = lo; b < hi; ++b) any |= lens[b] > 0;
        if (!any) return;
    }
    const int upb = (units + XT_NBLK - 1) / XT_NBLK;            // <= 8: two units per wave
    const int u0 = blk * upb;
    // this wave's rows of W: lane l holds columns 4 l + 256 it .. + 3 of each
    f32x4 wreg[2][XT_MAXIT];
    float wb[2];
#pragma unroll
    for (int uw = 0; uw < 2; ++uw) {
        const int uu = wave + 4 * uw, u = u0 + uu;
        const bool live = uu < upb && u < units;
        wb[uw] = (live && bias) ? bias[u] : 0.0f;
#pragma unroll
        for (int it = 0; it < XT_MAXIT; ++it) {
            const int k = lane * 4 + it * 256;
            wreg[uw][it] = (live && k < in_pad) ? *reinterpret_cast<const f32x4*>(W + (int64_t)u * ldw + k) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    // LDA rows of this slice's units and their global-mean entries: registers of thread c (column c of A)
    float areg[8], mreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int u = u0 + i;
        const bool live = i < upb && u < units && tid < out_dim;
        areg[i] = live ? A[(int64_t)u * out_dim + tid] : 0.0f;
        mreg[i] = (live && mean) ? mean[u] : 0.0f;
    }
    const int64_t b_lo = (int64_t)blockIdx.y * G, b_hi = b_lo + G < B ? b_lo + G : B;
    // pooled rows are prefetched one utterance ahead into registers (thread t: elements t + 256 j), so their global-load latency
    // hides under the previous utterance's arithmetic
    float xn[XT_MAXIT];
    if (pooled && b_lo < b_hi) {
#pragma unroll
        for (int j = 0; j < XT_MAXIT; ++j) {
            const int i = tid + 256 * j;
            xn[j] = i < in_dim ? pooled[b_lo * ld_pooled + i] : 0.0f;
        }
    }
    for (int64_t b = b_lo; b < b_hi; ++b) {
        // (1) the pooled vector (zeros beyond in_dim up to the 3072 the K loop covers)
        if (pooled) {
#pragma unroll
            for (int j = 0; j < XT_MAXIT; ++j) xs[tid + 256 * j] = xn[j];
            if (b + 1 < b_hi) {
#pragma unroll
                for (int j = 0; j < XT_MAXIT; ++j) {
                    const int i = tid + 256 * j;
                    xn[j] = i < in_dim ? pooled[(b + 1) * ld_pooled + i] : 0.0f;
                }
            }
            if (skip_empty && lens && lens[b] <= 0) continue;  // (workgroup-uniform; no barrier of this iteration has been reached)
        } else {
            const int len = lens ? lens[b] : (int)T;
            if (skip_empty && len <= 0) continue;              // (workgroup-uniform; no barrier of this iteration has been reached)
            const double n = (double)len;
            const int used = slots ? (len + slot_rows - 1) / slot_rows : 1;
            for (int c = tid; c < D; c += XT_THREADS) {
                double sv = 0.0, q = 0.0;
                if (slots == 0) {
                    sv = sums[(b * 2) * D + c];
                    q = sums[(b * 2 + 1) * D + c];
                } else {
                    for (int k = 0; k < used; ++k) {
                        sv += sums[((b * slots + k) * 2) * D + c];
                        q += sums[((b * slots + k) * 2 + 1) * D + c];
                    }
                }
                const double m = sv / n;
                xs[c] = (float)m;
                if (include_std) xs[D + c] = (float)sqrt(relu_keep_nan(q / n - m * m) + (double)eps);
            }
            for (int i = in_dim + tid; i < XT_MAXIT * 256; i += XT_THREADS) xs[i] = 0.0f;
        }
        __syncthreads();
        // (2) this workgroup's units of the affine: K order fixed per lane, one fixed butterfly
#pragma unroll
        for (int uw = 0; uw < 2; ++uw) {
            const int uu = wave + 4 * uw, u = u0 + uu;
            float acc = 0.0f;
#pragma unroll
            for (int it = 0; it < XT_MAXIT; ++it) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + lane * 4 + it * 256);
                acc = fmaf(wreg[uw][it].x, xv.x, acc);
                acc = fmaf(wreg[uw][it].y, xv.y, acc);
                acc = fmaf(wreg[uw][it].z, xv.z, acc);
                acc = fmaf(wreg[uw][it].w, xv.w, acc);
            }
            acc = wave_sum(acc);
            if (lane == 0 && uu < upb) {
                const float h = u < units ? acc + wb[uw] : 0.0f;
                hs[uu] = h;
                if (h_out && u < units) h_out[b * units + u] = h;
            }
        }
        __syncthreads();
        // (3) contribution of these units to every LDA output, into this slice's slot (hs is rewritten only behind the next
        // utterance's first barrier)
        if (tid < out_dim) {
            float pacc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < upb && u0 + i < units) pacc = fmaf(hs[i] - mreg[i], areg[i], pacc);
            partial[(b * XT_NBLK + blk) * out_dim + tid] = pacc;
        }
    }
    if (two_phase) return;                                 // the reduction is the next launch
    // (4) hand-off, ONCE per workgroup (an agent-scope release writes the XCD's dirty L2 lines back: microseconds): every wave's
    // stores drained -> one release -> one ticket per utterance of the group -> (if any was the last) one acquire
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) {                                       // the tickets of the group's utterances are drawn by parallel lanes (a returned
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // atomic is a memory round trip: 32 in a row cost more than the arithmetic)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int64_t bb = b_lo; bb < b_hi; bb += 64) {
            const int64_t b = bb + lane;
            unsigned last = 0;
            if (b < b_hi) {
                const unsigned t = __hip_atomic_fetch_add(counters + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = t == XT_NBLK - 1 ? 1u : 0u;
                if (last) __hip_atomic_store(counters + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
                xs[b - b_lo] = last ? 1.0f : 0.0f;
            }
            if (__ballot(last != 0)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    __syncthreads();
    for (int64_t b = b_lo; b < b_hi; ++b)
        if (xs[b - b_lo] != 0.0f) xt_reduce(partial, off, y, b, out_dim, ys, red);          // workgroup-uniform
}

// Second phase as a launch of its own (large batches: one agent-scope release per workgroup of the fused form writes the
// XCD's dirty L2 lines back, which with thousands of workgroups costs more than a kernel boundary). Same arithmetic.
__global__ __launch_bounds__(XT_THREADS) void xvec_tail_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ off,
                                                                      float* __restrict__ y, int out_dim, const int32_t* __restrict__ lens,
                                                                      int skip_empty) {
    extern __shared__ __attribute__((aligned(16))) float xr_sm[];
    if (skip_empty && lens && lens[blockIdx.x] <= 0) return;
    xt_reduce(partial, off, y, blockIdx.x, out_dim, xr_sm, xr_sm + out_dim);
}

// ------------------------------------------------------------------------------------ PLDA
template <typename R>
__device__ __forceinline__ R wsum(R v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename R>
__device__ __forceinline__ R rsqrt_(R v);
template <> __device__ __forceinline__ float rsqrt_<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double rsqrt_<double>(double v) { return sqrt(v); }
template <typename R>
__device__ __forceinline__ R rlog_(R v);
template <> __device__ __forceinline__ float rlog_<float>(float v) { return logf(v); }
template <> __device__ __forceinline__ double rlog_<double>(double v) { return log(v); }

// transformVector (plda.py:163-196): one workgroup per input vector; one wave per output row (strided).
template <typename R>
__global__ __launch_bounds__(256) void plda_transform_kernel(const R* __restrict__ x, int dim, const R* __restrict__ A,
                                                             const R* __restrict__ offset, const R* __restrict__ psi,
                                                             int normalize, int simple, R* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    R* xs = reinterpret_cast<R*>(smraw);
    R* ys = xs + dim;
    R* red = ys + dim;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < dim; i += 256) xs[i] = x[(int64_t)b * dim + i];
    __syncthreads();
    for (int r = wave; r < dim; r += 4) {
        R acc = 0;
        for (int c = lane; c < dim; c += 64) acc += A[(int64_t)r * dim + c] * xs[c];
        acc = wsum<R>(acc);
        if (lane == 0) ys[r] = acc + offset[r];
    }
    __syncthreads();
    R f = 1;
    if (normalize) {
        R part = 0;
        for (int r = threadIdx.x; r < dim; r += 256) {
            const R v = ys[r];
            part += simple ? v * v : v * v / (psi[r] + (R)1);
        }
        part = wsum<R>(part);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        const R tot = red[0] + red[1] + red[2] + red[3];
        f = simple ? rsqrt_<R>((R)dim) / rsqrt_<R>(tot) : rsqrt_<R>((R)dim / tot);
    }
    for (int r = threadIdx.x; r < dim; r += 256) out[(int64_t)b * dim + r] = ys[r] * f;
}

// logLikelihoodRatio (plda.py:198-245): a 64 x 64 block of (test i, class j) pairs per workgroup, sixteen pairs per thread (rows
// ti + 16 a, classes tj + 16 b), the dimensions staged through LDS 64 at a time.
// Per dimension the reference forms  mean = psi / (psi + 1) * y_j,  var1 = 1 + psi / (psi + 1),  var2 = 1 + psi  and sums
// (y_i - mean)^2 / var1 and y_i^2 / var2. Neither variance depends on the pair and the second sum not on j: a tile computes the
// per-dimension constants k = psi / (psi + 1), 1 / var1, 1 / var2 once (its only divisions), stages the class rows as k * y_j, sums
// y_i^2 / var2 once per row, and a pair costs one subtraction, one multiplication and one fused multiply-add per dimension, on operands
// that four pairs share (it was three fp64 divisions and three LDS reads per pair and dimension: 0.32 ms for 1024 x 1024 trials of
// dimension 128). Rows are padded by one element: the rows a wave reads side by side would otherwise sit in the same LDS banks.
#define PLDA_TILE 64
#define PLDA_DC 64
#define PLDA_LDS_BYTES(R) (sizeof(R) * (2 * PLDA_TILE * (PLDA_DC + 1) + 2 * PLDA_DC + 8))        // 67,648 B in fp64
template <typename R>
__global__ __launch_bounds__(256) void plda_score_kernel(const R* __restrict__ y, int64_t B, const R* __restrict__ yc,
                                                         int64_t Bc, int dim, const R* __restrict__ psi,
                                                         R* __restrict__ scores) {
    // rows i: vectors y (B of them, "test"); columns j: vectors yc (Bc of them, the classes); PLDA.call uses y == yc
    constexpr int LD = PLDA_DC + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];      // PLDA_LDS_BYTES(R)
    R* yi = reinterpret_cast<R*>(smraw);       // test rows, this chunk of dimensions
    R* yj = yi + PLDA_TILE * LD;               // class rows times k
    R* iv1 = yj + PLDA_TILE * LD;
    R* iv2 = iv1 + PLDA_DC;
    R* red = iv2 + PLDA_DC;                    // 8
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int64_t i0 = (int64_t)blockIdx.y * PLDA_TILE, j0 = (int64_t)blockIdx.x * PLDA_TILE;
    // constant terms: sum log(var1) and sum log(var2)
    R l1 = 0, l2 = 0;
    for (int d = threadIdx.x; d < dim; d += 256) {
        const R p = psi[d];
        l1 += rlog_<R>((R)1 + p / (p + (R)1));
        l2 += rlog_<R>((R)1 + p);
    }
    l1 = wsum<R>(l1); l2 = wsum<R>(l2);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = l1; red[4 + (threadIdx.x >> 6)] = l2; }
    R a[4][4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c[u] = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) a[u][v] = 0;
    }
    for (int d0 = 0; d0 < dim; d0 += PLDA_DC) {
        const int dc = min(PLDA_DC, dim - d0);
        __syncthreads();                                       // (the previous chunk has been consumed)
        for (int e = threadIdx.x; e < PLDA_TILE * PLDA_DC; e += 256) {
            const int r = e / PLDA_DC, dd = e - r * PLDA_DC;
            R vi = 0, vj = 0;
            if (dd < dc) {
                const R p = psi[d0 + dd];
                if (i0 + r < B) vi = y[(i0 + r) * dim + d0 + dd];
                if (j0 + r < Bc) vj = p * yc[(j0 + r) * dim + d0 + dd] / (p + (R)1);
            }
            yi[r * LD + dd] = vi;
            yj[r * LD + dd] = vj;
        }
        if (threadIdx.x < PLDA_DC) {
            const R p = threadIdx.x < dc ? psi[d0 + threadIdx.x] : (R)0;
            iv1[threadIdx.x] = (R)1 / ((R)1 + p / (p + (R)1));
            iv2[threadIdx.x] = (R)1 / ((R)1 + p);
        }
        __syncthreads();
        // sum_d y_i^2 / var2: the sixteen threads of a row group take every sixteenth dimension (added up across the lanes at the end)
        for (int dd = tj; dd < dc; dd += 16) {
            const R w2 = iv2[dd];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const R v = yi[(ti + 16 * u) * LD + dd];
                c[u] += v * v * w2;
            }
        }
        for (int dd = 0; dd < dc; ++dd) {
            const R w1 = iv1[dd];
            R vi[4], vj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vi[u] = yi[(ti + 16 * u) * LD + dd];
                vj[u] = yj[(tj + 16 * u) * LD + dd];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const R diff = vi[u] - vj[v];
                    a[u][v] += diff * diff * w1;
                }
        }
    }
    const R logdet1 = red[0] + red[1] + red[2] + red[3];       // (written before the first barrier of the chunk loop; dim >= 1)
    const R logdet2 = red[4] + red[5] + red[6] + red[7];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) c[u] += __shfl_xor(c[u], o, 64);
        const int64_t i = i0 + ti + 16 * u;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t j = j0 + tj + 16 * v;
            if (i < B && j < Bc) scores[i * Bc + j] = (R)(-0.5) * (logdet1 + a[u][v]) - (R)(-0.5) * (logdet2 + c[u]);
        }
    }
}

template <typename R>
static int plda_launch(const char* who, const R* x, int64_t B, int32_t dim, const R* A, const R* offset, const R* psi,
                       int32_t normalize_length, int32_t simple_length_norm, R* transformed, R* scores, void* stream) {
    KTF_REQUIRE(x && A && offset && psi && transformed, "%s: null argument", who);
    KTF_REQUIRE(B >= 0 && dim > 0, "%s: bad sizes", who);
    if (B == 0) return KTF_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds1 = sizeof(R) * (2 * (size_t)dim + 8);
    KTF_REQUIRE(lds1 <= 64 * 1024, "%s: dim %d too large", who, dim);
    hipLaunchKernelGGL(plda_transform_kernel<R>, dim3((unsigned)B), dim3(256), lds1, st, x, dim, A, offset, psi,
                       normalize_length, simple_length_norm, transformed);
    KTF_CHECK_LAUNCH(who);
    if (scores) {
        dim3 grid((unsigned)ktf_cdiv(B, PLDA_TILE), (unsigned)ktf_cdiv(B, PLDA_TILE));
        KTF_LDS_ONCE((int)PLDA_LDS_BYTES(R), plda_score_kernel<R>);
        hipLaunchKernelGGL(plda_score_kernel<R>, grid, dim3(256), PLDA_LDS_BYTES(R), st, transformed, B, transformed, B, dim, psi, scores);
        KTF_CHECK_LAUNCH(who);
    }
    return KTF_OK;
}

template <typename R>
static int plda_score_launch(const char* who, const R* test, int64_t N, const R* enroll, int64_t M, int32_t dim,
                             const R* psi, R* scores, void* stream) {
    KTF_REQUIRE(test && enroll && psi && scores, "%s: null argument", who);
    KTF_REQUIRE(N >= 0 && M >= 0 && dim > 0, "%s: bad sizes", who);
    if (N == 0 || M == 0) return KTF_OK;
    KTF_REQUIRE(ktf_cdiv(N, PLDA_TILE) < 65536, "%s: too many rows (shard them)", who);
    dim3 grid((unsigned)ktf_cdiv(M, PLDA_TILE), (unsigned)ktf_cdiv(N, PLDA_TILE));
    KTF_LDS_ONCE((int)PLDA_LDS_BYTES(R), plda_score_kernel<R>);
    hipLaunchKernelGGL(plda_score_kernel<R>, grid, dim3(256), PLDA_LDS_BYTES(R), (hipStream_t)stream, test, N, enroll, M, dim, psi, scores);
    KTF_CHECK_LAUNCH(who);
    return KTF_OK;
}

extern "C" int ktf_plda_score_f64(const double* test_tr, int64_t N, const double* enroll_tr, int64_t M, int32_t dim,
                                  const double* psi, double* scores, void* stream) {
    return plda_score_launch<double>("ktf_plda_score_f64", test_tr, N, enroll_tr, M, dim, psi, scores, stream);
}
extern "C" int ktf_plda_score_f32(const float* test_tr, int64_t N, const float* enroll_tr, int64_t M, int32_t dim,
                                  const float* psi, float* scores, void* stream) {
    return plda_score_launch<float>("ktf_plda_score_f32", test_tr, N, enroll_tr, M, dim, psi, scores, stream);
}

extern "C" int ktf_plda_f64(const double* x, int64_t B, int32_t dim, const double* A, const double* offset,
                            const double* psi, int32_t normalize_length, int32_t simple_length_norm,
                            double* transformed, double* scores, void* stream) {
    return plda_launch<double>("ktf_plda_f64", x, B, dim, A, offset, psi, normalize_length, simple_length_norm,
                               transformed, scores, stream);
}
extern "C" int ktf_plda_f32(const float* x, int64_t B, int32_t dim, const float* A, const float* offset,
                            const float* psi, int32_t normalize_length, int32_t simple_length_norm, float* transformed,
                            float* scores, void* stream) {
    return plda_launch<float>("ktf_plda_f32", x, B, dim, A, offset, psi, normalize_length, simple_length_norm,
                              transformed, scores, stream);
}

extern "C" int ktf_stats_pool(const void* x, int32_t x_dtype, int64_t B, int64_t T, int32_t D, int64_t ldx,
                              const int32_t* lens, int32_t input_period, int32_t include_std, float eps, float* out,
                              int64_t ld_out, void* stream) {
    KTF_REQUIRE(out && (x || T == 0), "ktf_stats_pool: null argument");      // (no frame at all: x may be an empty tensor; the means are 0 / 0 as in the reference)
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0 && ldx >= D, "ktf_stats_pool: bad sizes");
    KTF_REQUIRE(input_period > 0, "ktf_stats_pool: input_period must be > 0");
    KTF_REQUIRE(B < 65536, "ktf_stats_pool: B too large");
    KTF_REQUIRE(ld_out >= (include_std ? 2 : 1) * (int64_t)D, "ktf_stats_pool: ld_out too small");
    if (B == 0) return KTF_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (ldx % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) & 7) == 0);
    // few utterances: 32-column workgroups (4x the workgroups, same results bit for bit)
    const bool narrow = (int64_t)ktf_cdiv(D, 128) * B < 256;
#define SP_LAUNCH(TY, CW)                                                                                              \
    do {                                                                                                               \
        dim3 grid((unsigned)ktf_cdiv(D, CW), (unsigned)B);                                                             \
        if (vec) hipLaunchKernelGGL((stats_pool_kernel<TY, true, CW>), grid, dim3(SP_THREADS), 0, st, (const TY*)x, T, D, ldx, lens, input_period, include_std, eps, out, ld_out); \
        else hipLaunchKernelGGL((stats_pool_kernel<TY, false, CW>), grid, dim3(SP_THREADS), 0, st, (const TY*)x, T, D, ldx, lens, input_period, include_std, eps, out, ld_out); \
    } while (0)
#define SP_LAUNCH_T(TY) do { if (narrow) SP_LAUNCH(TY, 32); else SP_LAUNCH(TY, 128); } while (0)
    if (x_dtype == KTF_F32) SP_LAUNCH_T(float);
    else if (x_dtype == KTF_BF16) SP_LAUNCH_T(unsigned short);
    else KTF_REQUIRE(false, "ktf_stats_pool: bad dtype");
#undef SP_LAUNCH_T
#undef SP_LAUNCH
    KTF_CHECK_LAUNCH("ktf_stats_pool");
    return KTF_OK;
}

extern "C" int ktf_stats_pool_windowed_f32(const float* x, int64_t B, int64_t T, int32_t D, int32_t left, int32_t right,
                                           int32_t input_period, int32_t output_period, int32_t start, int64_t T_out,
                                           int32_t include_std, float eps, float* out, void* stream) {
    KTF_REQUIRE(x && out, "ktf_stats_pool_windowed_f32: null argument");
    KTF_REQUIRE(B >= 0 && T >= 0 && D > 0 && T_out >= 0, "ktf_stats_pool_windowed_f32: bad sizes");
    KTF_REQUIRE(left <= 0 && right >= 0, "ktf_stats_pool_windowed_f32: 'left_context' must be <= 0 and 'right_context' must be >= 0");
    KTF_REQUIRE(input_period > 0 && output_period > 0, "ktf_stats_pool_windowed_f32: periods must be > 0");
    const int64_t total = B * T_out * D;
    if (total == 0) return KTF_OK;
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stats_pool_windowed_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, B, T, D, left, right,
                       input_period, output_period, start, T_out, include_std, eps, out);
    KTF_CHECK_LAUNCH("ktf_stats_pool_windowed_f32");
    return KTF_OK;
}

extern "C" int ktf_xvec_tail_f32(const float* pooled, int64_t ld_pooled, const double* sums, int64_t slots, int32_t slot_rows, const int32_t* lens, int64_t T,
                                 int64_t B, int32_t D, int32_t include_std, float eps, const float* W, int64_t ldw, const float* bias,
                                 int32_t units, const float* mean, const float* A, const float* off, int32_t out_dim, float* partial,
                                 uint32_t* counters, float* y, float* h_out, int32_t group, int32_t flags, void* stream) {
    KTF_REQUIRE((pooled != nullptr) != (sums != nullptr), "ktf_xvec_tail_f32: exactly one of pooled / sums");
    KTF_REQUIRE(W && A && partial && counters && y, "ktf_xvec_tail_f32: null argument");
    const int in_dim = (include_std ? 2 : 1) * D;
    KTF_REQUIRE(B >= 0 && D > 0 && units > 0 && units <= 8 * XT_NBLK && in_dim <= XT_MAXIT * 256 && out_dim > 0 && out_dim <= XT_THREADS && group >= 1 && group <= 1024,
                "ktf_xvec_tail_f32: sizes outside the kernel (units <= 512, (1 + include_std) * D <= 3072, out_dim <= 256, 1 <= group <= 1024)");
    KTF_REQUIRE((B + group - 1) / group < 65536 && B < (1ll << 31), "ktf_xvec_tail_f32: too many utterance groups");
    KTF_REQUIRE(ldw >= ((in_dim + 3) & ~3) && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0, "ktf_xvec_tail_f32: W rows must be 16-byte aligned and padded to a multiple of 4 columns");
    if (pooled) KTF_REQUIRE(ld_pooled >= in_dim, "ktf_xvec_tail_f32: ld_pooled < in_dim");
    if (sums && slots) KTF_REQUIRE(slot_rows > 0 && slots * slot_rows >= T, "ktf_xvec_tail_f32: too few slots");
    if (B == 0) return KTF_OK;
    const size_t lds = sizeof(float) * ((size_t)XT_MAXIT * 256 + 16 + out_dim + 8);
    const int skip = (flags & KTF_TAIL_SKIP_EMPTY) ? 1 : 0;
    KTF_REQUIRE(!skip || lens, "ktf_xvec_tail_f32: KTF_TAIL_SKIP_EMPTY needs lens");
    const int two_phase = group > 1;        // many workgroups: the slot reduction as a second launch instead of in-kernel tickets
    hipLaunchKernelGGL(xvec_tail_kernel, dim3(XT_NBLK, (unsigned)((B + group - 1) / group)), dim3(XT_THREADS), lds, (hipStream_t)stream, pooled, ld_pooled,
                       sums, slots, (int)slot_rows, lens, T, (int)D, (int)include_std, eps, W, ldw, bias, in_dim, (int)units, mean, A, off, (int)out_dim, partial,
                       counters, y, h_out, B, (int)group, two_phase, skip);
    if (two_phase)
        hipLaunchKernelGGL(xvec_tail_reduce_kernel, dim3((unsigned)B), dim3(XT_THREADS), sizeof(float) * (out_dim + 8), (hipStream_t)stream, partial, off, y,
                           (int)out_dim, lens, skip);
    KTF_CHECK_LAUNCH("ktf_xvec_tail_f32");
    return KTF_OK;
}

extern "C" int ktf_xvec_post_f32(const float* x, int64_t B, int32_t in_dim, int32_t out_dim, const float* mean,
                                 const float* A, const float* off, float* y, void* stream) {
    KTF_REQUIRE(x && A && y, "ktf_xvec_post_f32: null argument");
    KTF_REQUIRE(B >= 0 && in_dim > 0 && out_dim > 0, "ktf_xvec_post_f32: bad sizes");
    if (B == 0) return KTF_OK;
    KTF_REQUIRE(sizeof(float) * ((size_t)in_dim + out_dim + XP_THREADS + 16) <= 64 * 1024, "ktf_xvec_post_f32: dims too large");
    // one embedding per workgroup while the batch fits one round of the chip; four from there on, if their rows fit the LDS
    const int eb = (B > 256 && sizeof(float) * 4 * ((size_t)in_dim + out_dim + XP_THREADS + 16) <= 64 * 1024) ? 4 : 1;
    const size_t lds = sizeof(float) * eb * ((size_t)in_dim + out_dim + XP_THREADS + 16);
    if (eb == 4)
        hipLaunchKernelGGL(xvec_post_kernel<4>, dim3((unsigned)((B + 3) / 4)), dim3(XP_THREADS), lds, (hipStream_t)stream, x, B, in_dim, out_dim, mean, A, off, y);
    else
        hipLaunchKernelGGL(xvec_post_kernel<1>, dim3((unsigned)B), dim3(XP_THREADS), lds, (hipStream_t)stream, x, B, in_dim, out_dim, mean, A, off, y);
    KTF_CHECK_LAUNCH("ktf_xvec_post_f32");
    return KTF_OK;
}
