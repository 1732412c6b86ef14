// Fused Kaldi-compatible feature front-end for gfx950.
//
// One wavefront (64 lanes) owns one frame: samples live in registers (lane l holds samples
// l, l+64, ...), DC / energy are wave reductions, pre-emphasis is a lane shuffle, the real FFT
// of size NF runs as a complex Stockham radix-4 FFT of size NF/2 in a wave-private LDS
// ping-pong, the mel bank is evaluated sparsely (each filter only over its own bins) and the
// DCT / lifter / C0<-energy epilogue writes the final cepstra. Frames are never materialised
// in HBM: algorithmic traffic is frame_shift*4 B in + num_ceps*4 B out per frame.
//
// Replaces: layers/dsp/framing.py:243-265, windowing.py:180-209, filterbank.py:225-242,
//           dct.py:175-176, mfcc.py:197-244 of the reference.
#include "common.h"

#define FE_WAVES 4
#define FE_THREADS (FE_WAVES * KTF_WAVE)
// Each wave works on wave-private LDS buffers: the LDS executes one wave's DS instructions in issue order, so a
// compiler-level ordering point is all a write -> cross-lane read hand-off inside ONE wave needs (no s_barrier).
#define WAVE_SYNC()                           \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_wave_barrier();      \
        asm volatile("" ::: "memory");        \
    } while (0)

struct FeLds {  // offsets (in floats) into the dynamic LDS block
    int window, tw, rtw, dct, lifter, mel_start, mel_len, mel_w, per_wave, wave_stride, total;
};

static FeLds fe_layout(const KtfFrontendCfg& c, const KtfFrontendTables& t, int out_stage) {
    FeLds L;
    int o = 0;
    const int n2 = c.nfft / 2;
    L.window = o; o += c.frame_size;
    L.tw = o; o += (out_stage >= KTF_OUT_FBANK) ? 2 * n2 : 0;
    L.rtw = o; o += (out_stage >= KTF_OUT_FBANK) ? 2 * n2 : 0;
    L.dct = o; o += (out_stage >= KTF_OUT_MFCC) ? c.num_mels * c.num_ceps : 0;
    L.lifter = o; o += (out_stage >= KTF_OUT_MFCC) ? c.num_ceps : 0;
    L.mel_start = o; o += (out_stage >= KTF_OUT_FBANK) ? c.num_mels : 0;
    L.mel_len = o; o += (out_stage >= KTF_OUT_FBANK) ? c.num_mels : 0;
    L.mel_w = o; o += (out_stage >= KTF_OUT_FBANK) ? c.num_mels * t.mel_stride : 0;
    o = (o + 3) & ~3;
    L.per_wave = o;
    L.wave_stride = (out_stage >= KTF_OUT_FBANK) ? 2 * c.nfft : 0;  // two complex buffers of NF/2
    L.total = o + FE_WAVES * L.wave_stride;
    return L;
}

// ---- counter-based Gaussian RNG for dither (Philox-4x32-10 + Box-Muller)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
__device__ __forceinline__ float gauss_noise(uint64_t seed, uint64_t row, uint32_t i) {
    uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32), i, 0x9E3779B9u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// One Stockham stage over N2 complex points held in LDS (x -> y); W[k] = exp(-2*pi*i*k/N2).
// n = current sub-transform size, s = N2/n = 1 << ls. Radix 4 while n % 4 == 0, else radix 2.
template <int N2>
__device__ __forceinline__ void fft_stage(const float2* __restrict__ x, float2* __restrict__ y,
                                          const float2* __restrict__ W, int n, int ls, int lane) {
    const int s = 1 << ls;
    if ((n & 3) == 0) {
        const int n1 = n >> 2;
        for (int bf = lane; bf < N2 / 4; bf += KTF_WAVE) {
            const int p = bf >> ls, q = bf & (s - 1);
            const float2 a = x[q + s * p], b = x[q + s * (p + n1)], c = x[q + s * (p + 2 * n1)],
                         d = x[q + s * (p + 3 * n1)];
            const float2 w1 = W[p * s], w2 = W[2 * p * s], w3 = W[3 * p * s];
            const float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
            const float2 bpd = make_float2(b.x + d.x, b.y + d.y);
            const float2 jbmd = make_float2(-(b.y - d.y), b.x - d.x);  // i*(b-d)
            y[q + s * (4 * p + 0)] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
            y[q + s * (4 * p + 1)] = cmul(w1, make_float2(amc.x - jbmd.x, amc.y - jbmd.y));
            y[q + s * (4 * p + 2)] = cmul(w2, make_float2(apc.x - bpd.x, apc.y - bpd.y));
            y[q + s * (4 * p + 3)] = cmul(w3, make_float2(amc.x + jbmd.x, amc.y + jbmd.y));
        }
    } else {
        const int m = n >> 1;
        for (int bf = lane; bf < N2 / 2; bf += KTF_WAVE) {
            const int p = bf >> ls, q = bf & (s - 1);
            const float2 a = x[q + s * p], b = x[q + s * (p + m)];
            y[q + s * (2 * p)] = make_float2(a.x + b.x, a.y + b.y);
            y[q + s * (2 * p + 1)] = cmul(W[p * s], make_float2(a.x - b.x, a.y - b.y));
        }
    }
}

template <int LOG2NF>
__global__ __launch_bounds__(FE_THREADS) void frontend_kernel(const void* __restrict__ in_v, int64_t B, int64_t n,
                                                              int in_kind, KtfFrontendCfg cfg, KtfFrontendTables tab,
                                                              FeLds L, int out_stage, float* __restrict__ out,
                                                              float* __restrict__ energy_out, uint64_t seed,
                                                              int64_t T) {
    constexpr int NF = 1 << LOG2NF;
    constexpr int N2 = NF / 2;
    constexpr int NV = NF / KTF_WAVE;  // samples per lane
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = cfg.frame_size;
    const int nm = cfg.num_mels, nc = cfg.num_ceps;
    const float* in = reinterpret_cast<const float*>(in_v);
    const int i16 = in_kind == KTF_IN_WAV_I16;
    if (i16) in_kind = KTF_IN_WAV;
    const int pad_left = (in_kind == KTF_IN_WAV && cfg.pad_mode) ? (M - cfg.frame_shift) / 2 : 0;
    const int64_t rstride = cfg.row_stride > 0 ? (int64_t)cfg.row_stride : n;

    // ---- block-shared tables -> LDS (once per block)
    if (in_kind != KTF_IN_WINDOWED && out_stage >= KTF_OUT_WINDOWED)
        for (int i = tid; i < M; i += FE_THREADS) lds[L.window + i] = tab.window[i];
    if (out_stage >= KTF_OUT_FBANK) {
        for (int i = tid; i < 2 * N2; i += FE_THREADS) {
            lds[L.tw + i] = tab.twiddle[i];
            lds[L.rtw + i] = tab.rtwiddle[i];
        }
        int* ms = reinterpret_cast<int*>(lds + L.mel_start);
        int* ml = reinterpret_cast<int*>(lds + L.mel_len);
        for (int i = tid; i < nm; i += FE_THREADS) {
            ms[i] = tab.mel_start[i];
            ml[i] = tab.mel_len[i];
        }
        for (int i = tid; i < nm * tab.mel_stride; i += FE_THREADS) lds[L.mel_w + i] = tab.mel_w[i];
    }
    if (out_stage >= KTF_OUT_MFCC) {
        for (int i = tid; i < nm * nc; i += FE_THREADS) lds[L.dct + i] = tab.dct[i];
        for (int i = tid; i < nc; i += FE_THREADS) lds[L.lifter + i] = (cfg.use_lifter && tab.lifter) ? tab.lifter[i] : 1.0f;
    }
    __syncthreads();

    float* bufA = lds + L.per_wave + wave * L.wave_stride;
    float* bufB = bufA + NF;
    const float2* W = reinterpret_cast<const float2*>(lds + L.tw);
    const float2* RW = reinterpret_cast<const float2*>(lds + L.rtw);
    const int* mel_start = reinterpret_cast<const int*>(lds + L.mel_start);
    const int* mel_len = reinterpret_cast<const int*>(lds + L.mel_len);

    const int64_t rows = B * T;
    const int64_t row_step = (int64_t)gridDim.x * FE_WAVES;
    for (int64_t row0 = (int64_t)blockIdx.x * FE_WAVES; row0 < rows; row0 += row_step) {
        const int64_t row = row0 + wave;
        const bool valid = row < rows;  // wave-uniform
        float v[NV];
        float logE = 0.0f;

        // ---- load the frame (Framing fused: frame t of utterance b starts at sample t*shift)
        if (valid && in_kind == KTF_IN_WAV && (i16 || cfg.pad_mode)) {
            // int16 samples and/or mirrored edges (KtfFrontendCfg.pad_mode): n < 2^31 checked by the host
            const int64_t b = row / T, t = row - b * T;
            const void* wav_b = i16 ? (const void*)(reinterpret_cast<const short*>(in_v) + b * rstride) : (const void*)(in + b * rstride);
            const int g0 = (int)t * cfg.frame_shift - pad_left;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + KTF_WAVE * j;
                v[j] = (i < M) ? ktf_wav_sample(wav_b, i16, (int)n, g0 + i) : 0.0f;
            }
        } else if (valid) {
            const float* src;
            if (in_kind == KTF_IN_WAV) {
                const int64_t b = row / T, t = row - b * T;
                src = in + b * rstride + t * (int64_t)cfg.frame_shift;
            } else {
                src = in + row * (int64_t)M;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + KTF_WAVE * j;
                v[j] = (i < M) ? src[i] : 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = 0.0f;
        }

        if (out_stage == KTF_OUT_FRAMES) {
            if (valid) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    if (i < M) out[row * (int64_t)M + i] = v[j];
                }
            }
            continue;
        }

        // ---- Windowing.call
        if (in_kind != KTF_IN_WINDOWED) {
            if (cfg.dither != 0.0f) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    if (i < M) v[j] += gauss_noise(seed, (uint64_t)row, (uint32_t)i) * cfg.dither;
                }
            }
            if (cfg.remove_dc) {
                double s = 0.0;   // fp64 accumulation: one rounding for the mean
#pragma unroll
                for (int j = 0; j < NV; ++j) s += (double)v[j];
                const float mean = (float)(wave_sum_d(s) / (double)M);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    if (i < M) v[j] -= mean;
                }
            }
            if (cfg.use_energy && cfg.raw_energy) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) s += (double)v[j] * (double)v[j];
                const float e = (float)log(fmax(wave_sum_d(s), 0.0) + (double)cfg.eps);
                logE = fmaxf(e, cfg.energy_floor);
            }
            if (cfg.preemph > 0.0f) {
                float y[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const float up = __shfl_up(v[j], 1, 64);                       // sample i-1 for lane >= 1
                    const float wrap = (j > 0) ? __shfl(v[j > 0 ? j - 1 : 0], 63, 64) : v[0];  // lane 0: sample i-1 (or x[0])
                    const float prev = (lane == 0) ? wrap : up;
                    y[j] = v[j] - cfg.preemph * prev;
                }
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    v[j] = (i < M) ? y[j] : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + KTF_WAVE * j;
                v[j] = (i < M) ? v[j] * lds[L.window + i] : 0.0f;
            }
            if (cfg.use_energy && !cfg.raw_energy) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) s += (double)v[j] * (double)v[j];
                const float e = (float)log(fmax(wave_sum_d(s), 0.0) + (double)cfg.eps);
                logE = fmaxf(e, cfg.energy_floor);
            }
        }

        if (out_stage == KTF_OUT_WINDOWED) {
            if (valid) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = lane + KTF_WAVE * j;
                    if (i < M) out[row * (int64_t)M + i] = v[j];
                }
                if (energy_out && cfg.use_energy && lane == 0) energy_out[row] = logE;
            }
            continue;
        }

        // ---- FilterBank.call: zero-padded real FFT of size NF via a complex FFT of size N2
#pragma unroll
        for (int j = 0; j < NV; ++j) bufA[lane + KTF_WAVE * j] = v[j];
        WAVE_SYNC();
        float2* src = reinterpret_cast<float2*>(bufA);
        float2* dst = reinterpret_cast<float2*>(bufB);
        {
            int nn = N2, ls = 0;
            while (nn > 1) {
                fft_stage<N2>(src, dst, W, nn, ls, lane);
                WAVE_SYNC();
                if ((nn & 3) == 0) { nn >>= 2; ls += 2; } else { nn >>= 1; ls += 1; }
                float2* t = src; src = dst; dst = t;
            }
        }
        // split the packed spectrum, take |X[k]| (^2); bin N2 carries no mel weight (filterbank.py:181)
        float* P = reinterpret_cast<float*>(dst);
        for (int k = lane; k < N2; k += KTF_WAVE) {
            const float2 zk = src[k], zm = src[(N2 - k) & (N2 - 1)];
            const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);
            const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);
            const float2 w = RW[k];
            const float xr = er + w.x * orr - w.y * oi;
            const float xi = ei + w.x * oi + w.y * orr;
            const float mag = sqrtf(xr * xr + xi * xi);
            P[k] = cfg.use_power ? mag * mag : mag;
        }
        WAVE_SYNC();
        float* feat = reinterpret_cast<float*>(src);  // FFT output no longer needed
        for (int f = lane; f < nm; f += KTF_WAVE) {
            const int s0 = mel_start[f], len = mel_len[f];
            const float* w = lds + L.mel_w + f * tab.mel_stride;
            float acc = 0.0f;
            for (int j = 0; j < len; ++j) acc += P[s0 + j] * w[j];
            if (cfg.use_log) acc = logf(fmaxf(acc, 0.0f) + cfg.eps);
            if (out_stage == KTF_OUT_FBANK) {
                if (valid) out[row * (int64_t)nm + f] = acc;
            } else {
                feat[f] = acc;
            }
        }
        WAVE_SYNC();
        if (out_stage == KTF_OUT_FBANK) continue;

        // ---- DCT.call + lifter + C0 <- log-energy (mfcc.py:205-228)
        for (int c = lane; c < nc; c += KTF_WAVE) {
            float acc = 0.0f;
            for (int m = 0; m < nm; ++m) acc += feat[m] * lds[L.dct + m * nc + c];
            acc *= lds[L.lifter + c];
            if (c == 0 && cfg.use_energy) acc = logE;
            if (valid) out[row * (int64_t)nc + c] = acc;
        }
        WAVE_SYNC();  // feat/P are rewritten by the next frame
    }
}

// Framing.call on its own (any frame size): out[b, t, i] = in[b, t*shift + i]
__global__ void framing_kernel(const void* __restrict__ in_v, int i16, int pad_left, int64_t B, int64_t n, int64_t rstride,
                               int M, int shift, int64_t T, float* __restrict__ out) {
    const int64_t total = B * T * M;
    const float* in = reinterpret_cast<const float*>(in_v);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % M);
        const int64_t bt = e / M;
        const int64_t t = bt % T, b = bt / T;
        if (i16 || pad_left >= 0) {        // pad_left < 0: plain reference Framing on fp32 samples
            const void* wav_b = i16 ? (const void*)(reinterpret_cast<const short*>(in_v) + b * rstride) : (const void*)(in + b * rstride);
            out[e] = ktf_wav_sample(wav_b, i16, (int)n, (int)t * shift - (pad_left < 0 ? 0 : pad_left) + i);
        } else {
            out[e] = in[b * rstride + t * shift + i];
        }
    }
}

// small dense row transform: out[r, c] = (sum_m x[r, m] * mat[m, c]) * scale[c]
__global__ void rowmat_kernel(const float* __restrict__ x, int64_t rows, int in_dim, int out_dim,
                              const float* __restrict__ mat, const float* __restrict__ scale,
                              float* __restrict__ out) {
    const int64_t total = rows * out_dim;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / out_dim;
        const int c = (int)(e - r * out_dim);
        float acc = 0.0f;
        for (int m = 0; m < in_dim; ++m) acc += x[r * in_dim + m] * mat[m * out_dim + c];
        out[e] = scale ? acc * scale[c] : acc;
    }
}

int ktf_frontend512_launch(const void* in, int64_t B, int64_t n, int32_t in_kind, const KtfFrontendCfg* cfg,
                           const KtfFrontendTables* tab, int32_t out_stage, float* out, uint64_t seed, int64_t T,
                           hipStream_t st);   // frontend512.hip

extern "C" int64_t ktf_num_frames(int64_t n_samples, int32_t frame_size, int32_t frame_shift) {
    if (frame_size <= 0 || frame_shift <= 0 || n_samples < frame_size) return 0;
    return 1 + (n_samples - frame_size) / frame_shift;
}

extern "C" int64_t ktf_num_frames_padded(int64_t n_samples, int32_t frame_size, int32_t frame_shift, int32_t pad_mode) {
    if (!pad_mode) return ktf_num_frames(n_samples, frame_size, frame_shift);
    if (frame_size <= 0 || frame_shift <= 0 || n_samples <= 0) return 0;
    // kaldi_numpy PadWaveform (frame_extraction.py:71-89): M frames, padded length Nv = (M-1)*shift + size
    const int64_t M = (n_samples + frame_shift / 2) / frame_shift;
    const int64_t Nv = (M - 1) * frame_shift + frame_size;
    const int64_t left = (frame_size - frame_shift) / 2, right = (Nv - n_samples) - left;
    if (M < 1 || Nv < n_samples || left < 0 || right < 0 || left > n_samples || right > n_samples) return -1;
    return M;
}

extern "C" int ktf_frontend_f32(const void* in, int64_t B, int64_t n, int32_t in_kind, const KtfFrontendCfg* cfg,
                                const KtfFrontendTables* tab, int32_t out_stage, float* out, float* energy,
                                uint64_t seed, void* stream) {
    KTF_REQUIRE(B >= 0 && n >= 0, "ktf_frontend_f32: negative size");
    KTF_REQUIRE(cfg && tab, "ktf_frontend_f32: null argument");
    if (B == 0) return KTF_OK;                       // empty batch: nothing to read or write (pointers may be NULL)
    KTF_REQUIRE(in && out, "ktf_frontend_f32: null argument");
    KTF_REQUIRE(in_kind >= KTF_IN_WAV && in_kind <= KTF_IN_WAV_I16, "ktf_frontend_f32: bad in_kind %d", in_kind);
    const bool wav = in_kind == KTF_IN_WAV || in_kind == KTF_IN_WAV_I16;
    const int pad_mode = wav ? cfg->pad_mode : 0;
    KTF_REQUIRE(pad_mode == 0 || pad_mode == 1, "ktf_frontend_f32: bad pad_mode %d", pad_mode);
    KTF_REQUIRE(cfg->row_stride >= 0, "ktf_frontend_f32: negative row_stride");
    if (wav && (pad_mode || in_kind == KTF_IN_WAV_I16)) KTF_REQUIRE(n < (1ll << 31) - 4096, "ktf_frontend_f32: int16 / padded input needs n < 2^31");
    int64_t Tw = 0;
    if (wav) {
        Tw = ktf_num_frames_padded(n, cfg->frame_size, cfg->frame_shift, pad_mode);
        KTF_REQUIRE(Tw >= 0, "ktf_frontend_f32: mirror padding undefined for %lld samples (frame %d, shift %d)", (long long)n, cfg->frame_size, cfg->frame_shift);
    }
    KTF_REQUIRE(out_stage >= KTF_OUT_FRAMES && out_stage <= KTF_OUT_MFCC, "ktf_frontend_f32: bad out_stage %d", out_stage);
    KTF_REQUIRE(cfg->frame_size > 0 && cfg->frame_shift > 0, "ktf_frontend_f32: frame size/shift must be > 0");
    KTF_REQUIRE(!(in_kind == KTF_IN_WINDOWED && out_stage < KTF_OUT_FBANK), "ktf_frontend_f32: windowed input needs a FBANK/MFCC stage");
    if (out_stage == KTF_OUT_FRAMES) {
        KTF_REQUIRE(wav, "ktf_frontend_f32: KTF_OUT_FRAMES needs KTF_IN_WAV");
        KTF_REQUIRE(B == 0 || pad_mode || n >= cfg->frame_size, "ktf_frontend_f32: input of %lld samples is shorter than a frame (%d)", (long long)n, cfg->frame_size);
        const int64_t Tf = Tw;
        const int64_t total = B * Tf * cfg->frame_size;
        if (total == 0) return KTF_OK;
        int blk = ktf_cdiv(total, 256);
        if (blk > 4096) blk = 4096;
        hipLaunchKernelGGL(framing_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, in, (int)(in_kind == KTF_IN_WAV_I16),
                           pad_mode ? (cfg->frame_size - cfg->frame_shift) / 2 : (in_kind == KTF_IN_WAV_I16 ? 0 : -1), B, n,
                           cfg->row_stride > 0 ? (int64_t)cfg->row_stride : n, cfg->frame_size, cfg->frame_shift, Tf, out);
        KTF_CHECK_LAUNCH("ktf_frontend_f32(framing)");
        return KTF_OK;
    }
    int log2nf = 0;
    while ((1 << log2nf) < cfg->nfft) ++log2nf;
    KTF_REQUIRE((1 << log2nf) == cfg->nfft && log2nf >= 6 && log2nf <= 11, "ktf_frontend_f32: nfft %d must be a power of two in [64, 2048]", cfg->nfft);
    KTF_REQUIRE(cfg->frame_size <= cfg->nfft, "ktf_frontend_f32: frame_size %d > nfft %d", cfg->frame_size, cfg->nfft);
    if (out_stage >= KTF_OUT_FBANK) {
        KTF_REQUIRE(cfg->num_mels >= 1 && cfg->num_mels <= 128, "ktf_frontend_f32: num_mels %d out of range", cfg->num_mels);
        KTF_REQUIRE(tab->twiddle && tab->rtwiddle && tab->mel_start && tab->mel_len && tab->mel_w && tab->mel_stride > 0, "ktf_frontend_f32: missing FFT/mel tables");
    }
    if (out_stage >= KTF_OUT_MFCC) {
        KTF_REQUIRE(cfg->num_ceps >= 1 && cfg->num_ceps <= cfg->num_mels, "ktf_frontend_f32: num_ceps %d must be in [1, num_mels]", cfg->num_ceps);
        KTF_REQUIRE(tab->dct, "ktf_frontend_f32: missing DCT table");
    }
    if (in_kind != KTF_IN_WINDOWED && out_stage >= KTF_OUT_WINDOWED) KTF_REQUIRE(tab->window, "ktf_frontend_f32: missing window table");
    const int64_t T = wav ? Tw : n;
    if (wav) KTF_REQUIRE(B == 0 || pad_mode || n >= cfg->frame_size, "ktf_frontend_f32: input of %lld samples is shorter than a frame (%d)", (long long)n, cfg->frame_size);
    const int64_t rows = B * T;
    if (rows == 0) return KTF_OK;
    if (cfg->nfft == 512 && out_stage >= KTF_OUT_FBANK && tab->fast_tw && tab->fast_mel_meta && tab->fast_mel_w &&
        cfg->num_mels <= 32 && (out_stage == KTF_OUT_FBANK || cfg->num_ceps <= 64) && B < 65536 &&
        T * (int64_t)512 < (1ll << 31) && n < (1ll << 31))
        return ktf_frontend512_launch(in, B, n, in_kind, cfg, tab, out_stage, out, seed, T, (hipStream_t)stream);
    FeLds L = fe_layout(*cfg, *tab, out_stage);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    KTF_REQUIRE(lds_bytes <= 160 * 1024, "ktf_frontend_f32: configuration needs %zu B of LDS (> 160 KiB)", lds_bytes);
    int blocks = ktf_cdiv(rows, FE_WAVES);
    const int max_blocks = 256 * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    hipStream_t st = (hipStream_t)stream;
#define FE_LAUNCH(LG)                                                                                          \
    case LG:                                                                                                   \
        if (lds_bytes > 64 * 1024)                                                                             \
            (void)hipFuncSetAttribute((const void*)frontend_kernel<LG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(frontend_kernel<LG>, dim3(blocks), dim3(FE_THREADS), lds_bytes, st, in, B, n, in_kind, \
                           *cfg, *tab, L, out_stage, out, energy, seed, T);                                     \
        break;
    switch (log2nf) {
        FE_LAUNCH(6) FE_LAUNCH(7) FE_LAUNCH(8) FE_LAUNCH(9) FE_LAUNCH(10) FE_LAUNCH(11)
        default: break;
    }
#undef FE_LAUNCH
    KTF_CHECK_LAUNCH("ktf_frontend_f32");
    return KTF_OK;
}

extern "C" int ktf_dct_f32(const float* x, int64_t rows, int32_t in_dim, int32_t out_dim, const float* dct,
                           const float* lifter, float* out, void* stream) {
    KTF_REQUIRE(x && dct && out, "ktf_dct_f32: null argument");
    KTF_REQUIRE(rows >= 0 && in_dim > 0 && out_dim > 0, "ktf_dct_f32: bad sizes");
    if (rows == 0) return KTF_OK;
    const int64_t total = rows * out_dim;
    int blocks = ktf_cdiv(total, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rowmat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, in_dim, out_dim, dct,
                       lifter, out);
    KTF_CHECK_LAUNCH("ktf_dct_f32");
    return KTF_OK;
}
