/*
 * ktf_hip.h — C-ABI of libktf_hip.so: the MI355X (gfx950) kernels behind the
 * wav -> x-vector hot path of shahruk10/kaldi-tflite.
 *
 * The reference has no FFI of its own: its operator API is the Keras layer protocol
 * and every layer's `call` is a chain of TensorFlow ops. Each entry point below
 * replaces the TF op chain of one (or a fused run of) reference layer call(s); the
 * reference location it replaces is cited per function (paths relative to
 * kaldi_tflite/lib/ in the reference tree). The Python host side
 * (kaldi-tflite_amd/kaldi_tflite_amd) mirrors the reference's ktf.layers / ktf.models
 * surface and binds these symbols with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain C symbols, plain pointers and sizes, no C++/torch types;
 *  - every pointer is a DEVICE pointer unless stated; the caller owns every buffer
 *    (including workspaces); the library allocates nothing and keeps no state except a
 *    thread-local error string;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream) and is safe to capture into a hipGraph;
 *  - return value: 0 = KTF_OK, negative = error (ktf_last_error() has the text);
 *  - fp32 tensors are row-major; "ld*" arguments are row strides in ELEMENTS;
 *  - ragged batches: activations are kept utterance-strided (B, T_max, D) with a device
 *    int32 `lens[B]` giving the number of valid rows of each utterance (NULL = all T).
 */
#ifndef KTF_HIP_H_
#define KTF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KTF_OK 0
#define KTF_EINVAL (-1)     /* bad argument */
#define KTF_ELAUNCH (-2)    /* HIP launch/runtime error */
#define KTF_EUNSUPPORTED (-3)

/* element types of activation / weight buffers */
#define KTF_F32 0
#define KTF_BF16 1
/* (2 was KTF_F16, the element type of the half-precision modes below) */
#define KTF_BF16P 3         /* a bf16 PAIR in an fp32-sized slot: bits 0-15 = bf16(v) (round to nearest even), bits 16-31 =
                             * bf16(v - bf16(v)): 16 mantissa bits, the shapes / strides / padding of fp32 (zero = 0x00000000).
                             * Operands of KTF_GEMM_BF16X4; KTF_GEMM_F32 and KTF_GEMM_BF16X4 can write it (y_dtype) */

/* GEMM arithmetic of ktf_tdnn */
#define KTF_GEMM_F32 0      /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (parity path) */
#define KTF_GEMM_BF16 1     /* v_mfma_f32_*_bf16: bf16 operands, fp32 accumulate */
#define KTF_GEMM_BF16X3 2   /* split-bf16: x=hi+lo, w=hi+lo, 3 bf16 MFMA passes, fp32 accumulate */
/* (3 and 4 were KTF_GEMM_F16, one half-precision pass, and KTF_GEMM_F16X2, two half passes with a calibrated one-pass tail: round 2's
 * timed mode, superseded by KTF_GEMM_F16MX -- faster, tighter on speech, no calibration. Removed from the library in round 5;
 * tools/mx/experiments/README.md names the commit at which they were last part of it.) */
#define KTF_GEMM_F16MX 5    /* ONE half-precision MFMA pass plus two block-scaled (OCP MX) residual passes on
                             * v_mfma_scale_f32_16x16x128_f8f6f4, which runs fp4 / fp6 operands at four times the half rate:
                             *   y = x_h * w_h + x_l4 * w_4 + x_4 * w_l6
                             * (x_h, w_h half; x_l4 = e2m1 image of x - x_h; w_4 = e2m1 image of w; x_4 = e2m1 image of x_h;
                             * w_l6 = e2m3 image of w - w_h; one E8M0 scale per 32 K elements). 1.5 MFMA passes per algorithmic
                             * flop with ~15 significant bits on BOTH operands, no calibration. What is left is zero-mean rounding
                             * noise per frame that the statistics pooling averages, so the max-abs x-vector deviation depends on
                             * the voiced length: 1.2e-5 on 10 s of noise, 2-5e-5 on 10 s of speech, 4-6.5e-5 on 5 s, up to 1.2e-4
                             * on 1-1.5 s (tests/test_gpu_margin.py). The host side (Sequential.MIN_FRAMES, XvectorExtractor.route_short_utterances) sends
                             * utterances below 400 voiced frames through KTF_GEMM_BF16X3. Through ktf_tdnn_mx / ktf_tdnn_mx_stats
                             * on the four-plane activation format ktf_mx_planes produces */
#define KTF_GEMM_BF16X4 6   /* all four bf16 products of (x_hi + x_lo)(w_hi + w_lo), fp32 accumulate, on SMALL tiles (64 x 32..96,
                             * csrc/tdnn_pair.hip): x and w of KTF_BF16P, y KTF_F32 or KTF_BF16P, through ktf_tdnn (w_lo NULL). For
                             * batches too small to fill the chip on 256-row tiles -- a single utterance -- where it replaces the
                             * fp32 small-tile kernels in every mode but KTF_GEMM_F32: x-vectors ~1e-5 from the fp64 oracle */

/* activations fused into the ktf_tdnn epilogue */
#define KTF_ACT_NONE 0
#define KTF_ACT_RELU 1
#define KTF_ACT_SIGMOID 2
#define KTF_ACT_TANH 3
/* the rest of tf.keras.activations of the reference's TensorFlow (2.8; layers/tdnn/tdnn.py:117-118 accepts any of its names):
 * ktf_tdnn with KTF_GEMM_F32 and an fp32 output runs them as a second launch over the rows it wrote (ktf_activation_f32); the 16-bit
 * and MX kernels fuse KTF_ACT_NONE / KTF_ACT_RELU only (the host runs such layers on the fp32 kernels) */
#define KTF_ACT_ELU 4            /* x > 0 ? x : exp(x) - 1 */
#define KTF_ACT_SELU 5           /* 1.0507 * (x > 0 ? x : 1.67326 * (exp(x) - 1)) */
#define KTF_ACT_SOFTPLUS 6       /* log(exp(x) + 1) */
#define KTF_ACT_SOFTSIGN 7       /* x / (|x| + 1) */
#define KTF_ACT_SWISH 8          /* x * sigmoid(x) */
#define KTF_ACT_GELU 9           /* 0.5 x (1 + erf(x / sqrt 2)): approximate=False, the default */
#define KTF_ACT_EXPONENTIAL 10   /* exp(x) */
#define KTF_ACT_HARD_SIGMOID 11  /* clip(0.2 x + 0.5, 0, 1) */
#define KTF_ACT_SOFTMAX 12       /* over the units of a row (axis = -1) */

int32_t ktf_version(void);
/* copies the calling thread's last error text (NUL-terminated) into buf; returns its length */
size_t ktf_last_error(char* buf, size_t cap);
/* 16 hex digits of the sha256 over the library's sources (csrc/Makefile): measurements kept under profiles/ name the build they were
 * made on with it (bench.py attaches a stored HBM-traffic figure only to the build it belongs to). A static string. */
const char* ktf_build_id(void);
/* measurement aid (bench.py): ONE wave that stays resident for `us` microseconds beside whatever else runs on `stream`'s device
 * and reads the shader clock (s_memtime) against the constant 100 MHz counter (s_memrealtime). out (device, 4 x uint64):
 * [0] shader clocks and [1] 100 MHz ticks over the whole stay, [2] / [3] the lowest / highest clock in kHz over ~1 ms windows.
 * Launch it on a stream of its own next to the timed work. 0 < us <= 10 000 000. */
int ktf_clock_probe(unsigned long long* out, int64_t us, void* stream);

/* ------------------------------------------------------------------ front-end (a1-a5)
 * Framing   layers/dsp/framing.py:243-265       (tf.gather of frame indexes)
 * Windowing layers/dsp/windowing.py:180-209     (dither, DC removal, log-energy, pre-emphasis, window)
 * FilterBank layers/dsp/filterbank.py:225-242   (pad, tf.signal.rfft, abs, pow, matmul mel, log)
 * DCT       layers/dsp/dct.py:175-176           (matmul)
 * MFCC      layers/dsp/mfcc.py:197-244          (the three above + lifter + C0 <- energy)
 */
typedef struct KtfFrontendCfg {
    int32_t frame_size;    /* samples per frame (= 2*(size//2), framing.py:104-106)            */
    int32_t frame_shift;   /* samples between frame starts                                      */
    int32_t nfft;          /* power of two >= frame_size, 64..2048 (filterbank.py:156-157)       */
    int32_t num_mels;      /* mel bins  (<= 128)                                                 */
    int32_t num_ceps;      /* cepstra kept (<= num_mels)                                         */
    int32_t remove_dc;     /* windowing.py:186-189                                               */
    int32_t raw_energy;    /* energy before (1) or after (0) pre-emphasis+window                 */
    int32_t use_energy;    /* compute log-energy (Windowing.return_energy / MFCC.use_energy)     */
    int32_t use_power;     /* |X|^2 (1) or |X| (0)                                               */
    int32_t use_log;       /* log(max(.,0)+eps) after the mel bank                               */
    int32_t use_lifter;    /* multiply cepstra by `lifter` (mfcc.py:211-212)                     */
    float preemph;         /* 0 disables                                                         */
    float dither;          /* 0 disables; else x += N(0,1)*dither (counter-based RNG, `seed`)    */
    float energy_floor;    /* clip of the LOG energy from below (windowing.py:177)               */
    float eps;             /* epsilon inside both logs                                           */
    int32_t pad_mode;      /* KTF_IN_WAV* only. 0: Framing as the reference (no padding, T = 1+(n-size)/shift).
                            * 1: Kaldi snip-edges=false framing: the waveform is mirror-padded as by the reference's
                            *    kaldi_numpy PadWaveform (kaldi_numpy/frame_extraction.py:28-89) -- fused into the frame
                            *    gather, nothing is materialised; T = (n + shift/2) / shift                             */
    int32_t row_stride;    /* KTF_IN_WAV* only: samples between the starts of consecutive rows of `in`; 0 = n (dense).
                            * A stride < n describes OVERLAPPING windows of one long recording (sliding-window /
                            * diarization extraction, torch `wav.unfold(-1, n, hop)`) without copying them            */
} KtfFrontendCfg;

typedef struct KtfFrontendTables {   /* all DEVICE pointers, built once by the host */
    const float* window;       /* [frame_size]                                             */
    const float* twiddle;      /* [nfft]   : (cos,-sin)(2*pi*k/(nfft/2)), k < nfft/2        */
    const float* rtwiddle;     /* [nfft]   : (cos,-sin)(2*pi*k/nfft),     k < nfft/2        */
    const int32_t* mel_start;  /* [num_mels] first FFT bin of each filter                  */
    const int32_t* mel_len;    /* [num_mels] number of bins of each filter                 */
    const float* mel_w;        /* [num_mels][mel_stride] weights from mel_start            */
    const float* dct;          /* [num_mels][num_ceps] (filterbank.py melBank / dct.py dct) */
    const float* lifter;       /* [num_ceps] or NULL                                       */
    /* optional tables of the register-resident nfft = 512 fast path (all three NULL = generic kernel); `reserved` = bins per mel work item */
    const float* fast_tw;      /* [64][18] per-lane FFT twiddles: (re,im) of W256^(n0 r), W64^(n1 r), W16^(n2 r), r=1..3,
                                  n0 = (l>>1)+32(l&1), n1 = (l>>1)&15, n2 = (l>>1)&3                                   */
    const int32_t* fast_mel_meta; /* [64][4] per-lane mel work item: first bin, bins (<=16), filter (-1 = idle),
                                     flags (1: lane+1 same filter, 2: lane+2 same filter, 4: first lane of the filter) */
    const float* fast_mel_w;   /* [64][16] weights of the work item                                                  */
    int32_t mel_stride;
    int32_t reserved;
} KtfFrontendTables;

/* input kinds / output stages of ktf_frontend_f32 */
#define KTF_IN_WAV 0         /* in = (B, N) samples; frames are gathered on the fly (Framing fused)   */
#define KTF_IN_FRAMES 1      /* in = (B, T, frame_size) frames                                        */
#define KTF_IN_WINDOWED 2    /* in = (B, T, frame_size) already-windowed frames (FilterBank alone)     */
#define KTF_IN_WAV_I16 3     /* in = (B, N) int16 PCM samples (half the HBM / PCIe bytes of KTF_IN_WAV) */
#define KTF_OUT_FRAMES 0     /* out (B,T,frame_size): Framing.call                                    */
#define KTF_OUT_WINDOWED 1   /* out (B,T,frame_size) [+ energy (B,T)]: Windowing.call                 */
#define KTF_OUT_FBANK 2      /* out (B,T,num_mels): FilterBank.call                                    */
#define KTF_OUT_MFCC 3       /* out (B,T,num_ceps): MFCC.call                                          */

/* Number of frames Framing produces from n samples: 1 + (n - frame_size) / frame_shift (0 if n < size). */
int64_t ktf_num_frames(int64_t n_samples, int32_t frame_size, int32_t frame_shift);
/* The same for a given KtfFrontendCfg.pad_mode (1: (n + shift/2) / shift; -1 if the mirror padding is undefined for n). */
int64_t ktf_num_frames_padded(int64_t n_samples, int32_t frame_size, int32_t frame_shift, int32_t pad_mode);

/* One launch for any prefix/suffix of Framing -> Windowing -> FilterBank -> DCT/lifter/C0.
 * `n` is the number of samples per row for KTF_IN_WAV, else the number of frames T.
 * `energy` (B,T) may be NULL unless out_stage == KTF_OUT_WINDOWED and cfg->use_energy. */
int ktf_frontend_f32(const void* in, int64_t B, int64_t n, int32_t in_kind, const KtfFrontendCfg* cfg,
                     const KtfFrontendTables* tab, int32_t out_stage, float* out, float* energy,
                     uint64_t seed, void* stream);

/* DCT.call (layers/dsp/dct.py:175-176) on its own: out[r, c] = sum_m x[r, m] * dct[m, c] (* lifter[c]). */
int ktf_dct_f32(const float* x, int64_t rows, int32_t in_dim, int32_t out_dim, const float* dct,
                const float* lifter, float* out, void* stream);

/* ------------------------------------------------------------------ VAD / compaction / CMVN (a6-a8)
 * VAD.call            layers/dsp/vad.py:156-203
 * gather_nd+expand    models/kaldi/xvector_extractor.py:163-165 (per utterance; see DESIGN.md on batch>1)
 * CMVN.call           layers/normalization/cmvn.py:186-250
 */
typedef struct KtfVadCfg {
    float energy_threshold;
    float energy_mean_scale;    /* 0 disables the mean term */
    float proportion_threshold;
    int32_t frames_context;
    int32_t energy_coeff;       /* column of the log-energy */
} KtfVadCfg;

typedef struct KtfCmvnCfg {
    int32_t window;      /* N */
    int32_t norm_vars;   /* divide by the windowed std (no epsilon, as the reference) */
    int32_t valid;       /* padding == "VALID": keep frames [N/2, T-(N-1)/2) when T > N */
    int32_t reserved;
} KtfCmvnCfg;

/* mask (B,T) fp32 of kept frames (VAD.call with return_indexes=False). */
int ktf_vad_mask_f32(const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* cfg, float* mask,
                     void* stream);
/* per-utterance compaction: idx (B,T) int32 = kept frame numbers in order, lens[B] = their count
 * (VAD.call with return_indexes=True gives the same rows as [b, idx[b, j]], j < lens[b]). */
int ktf_vad_index(const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* cfg, int32_t* idx,
                  int32_t* lens, void* stream);
/* CMVN of each utterance's first lens[b] rows (lens NULL = T rows). x (B,T,D) rows of stride ldx;
 * out rows of stride ldo >= D, columns D..ldo-1 are written as zeros. `work` = B*T*2*D floats.
 * With cfg->valid the output row j holds input frame j + N/2 and out_lens[b] (may be NULL) gets the count. */
int ktf_cmvn_f32(const float* x, int64_t B, int64_t T, int32_t D, int64_t ldx, const int32_t* lens,
                 const KtfCmvnCfg* cfg, float* out, int64_t ldo, int32_t* out_lens, float* work, void* stream);
/* Fused hot path: VAD -> per-utterance compaction -> CMVN (xvector_extractor.py:162-166).
 * out_dtype KTF_F32 or KTF_BF16. idx_work = B*T int32 (on return: the kept frame numbers of each utterance),
 * work = B*T*2*D floats (touched only by recordings too long for the LDS: more than ~38,000 frames). Any T < 2^31 / ldo.
 * Batches of fewer than 256 utterances spread each utterance over up to eight workgroups (same values, bit for bit). */
int ktf_vad_cmvn(const float* feats, int64_t B, int64_t T, int32_t D, const KtfVadCfg* vad, const KtfCmvnCfg* cmvn,
                 void* out, int32_t out_dtype, int64_t ldo, int32_t* lens, int32_t* idx_work, float* work,
                 void* stream);

/* Per-utterance routing by voiced length (no reference counterpart: the reference runs one utterance at a time in fp32). lens (B) ->
 * lens_main[b] = lens[b] >= min_frames ? lens[b] : 0 and lens_short[b] = 0 < lens[b] < min_frames ? lens[b] : 0. host_flag (optional):
 * two int32 in PINNED, device-visible host memory: [0] = number of short utterances, then [1] = seq (system-scope release), for a host
 * that polls [1] instead of synchronising the stream. */
int ktf_route_short(const int32_t* lens, int64_t B, int32_t min_frames, int32_t* lens_main, int32_t* lens_short, int32_t* host_flag,
                    int32_t seq, void* stream);

/* ------------------------------------------------------------------ TDNN stack (a9, a10)
 * TDNN.call   layers/tdnn/tdnn.py:251-280 (gather im2col + conv2d 1xK + bias + activation)
 * ReLU / BatchNorm (inference affine)  models/kaldi/sequential.py:72-74, layers/normalization/batchnorm.py:78-88
 *
 * y[b, t, u] = post( act( bias[u] + sum_k sum_d x[b, row(t,k), d] * W[u, k*Din_pad + d] ) )
 *   row(t,k)  = clip(start + t*subsampling + ctx[k], 0, len_b-1)  ("SAME": replicate edges; "VALID": no clip needed)
 *   post(v)   = v * scale[u] + shift[u]   when scale != NULL (BatchNorm folded to an affine)
 * x: (B, T, ldx) of x_dtype; W: (units_pad, nctx*Din_pad) row-major of w_dtype, zero padded, where Din_pad is
 * Din rounded up to a multiple of 32 (must be <= ldx; pad columns of x must be finite) and units_pad is units rounded
 * up to 256 (the widest N-tile of the kernels). For KTF_GEMM_BF16X3 `w` holds the hi part and `w_lo` the lo part (both bf16); x is fp32.
 * y: (B, T_out_max, ldy) of y_dtype; out_lens[b] (may be NULL) receives the valid output rows of utterance b.
 */
typedef struct KtfTdnnDesc {
    int32_t units;
    int32_t din;            /* logical input feature dim */
    int32_t din_pad;        /* multiple of 32, <= ldx     */
    int32_t nctx;           /* <= 16 */
    int32_t ctx[16];        /* sorted ascending */
    int32_t subsampling;
    int32_t valid;          /* padding == "VALID" */
    int32_t act;            /* KTF_ACT_* */
    int32_t gemm;           /* KTF_GEMM_* */
    int32_t x_dtype, w_dtype, y_dtype;
    int32_t flags;          /* KTF_TDNN_* bits, 0 = default */
} KtfTdnnDesc;

/* KtfTdnnDesc.flags */
#define KTF_TDNN_REF_TILES 1      /* KTF_GEMM_F32 only: run the register-staged 32x32x2 tile kernels, the bitwise reference the
                                   * LDS-DMA-staged fp32 kernels are tested against (slower; same bits) */
#define KTF_TDNN_DET_STATS 2      /* ktf_tdnn_stats / ktf_tdnn_split_stats only: run-to-run reproducible pooling. Every
                                   * 128-row block of an utterance stores its fp64 column sums in a slot of its own instead
                                   * of adding them with atomics; `sums` is then (B, ktf_stats_slots(T), 2, units), need not
                                   * be zeroed, and is reduced in slot order by ktf_stats_finalize_slots */
#define KTF_TDNN_K_INTERLEAVED 4  /* ktf_tdnn_split / ktf_tdnn_split_stats only: the K axis of W (both planes) is ordered
                                   * (32-feature chunk, context, feature in chunk) instead of (context, feature), i.e. column
                                   * ((d / 32) * nctx + k) * 32 + d % 32 holds W[u, k * Din_pad + d]. The kernel then walks
                                   * the contexts of one feature chunk in consecutive K-steps, so the three (five) reads of an
                                   * activation row piece by a multi-context layer are adjacent in time and hit in L2 instead of
                                   * returning to HBM / MALL 16 K-steps apart */
#define KTF_TDNN_W_TILED 8        /* ktf_tdnn_split / ktf_tdnn_split_stats only: W (both planes) is stored as the kernel's LDS
                                   * stage images instead of row-major: for N-tile nt (256 units) and K-step ks (32 columns of
                                   * the K order in force) one contiguous 16 KiB block at ((nt * ktot / 32 + ks) * 16 KiB),
                                   * holding for row r and 16-byte position q the columns 32 ks + 8 (q ^ ((4 - (r >> 2)) & 3))
                                   * .. + 7 of unit 256 nt + r at byte 64 r + 16 q. A weight DMA instruction then copies 1 KiB of
                                   * consecutive bytes (8 whole cache lines) instead of gathering 16 rows x 64 B */
/* (16, 32 and bits 8..23 were KTF_TDNN_X_CHUNKED / KTF_TDNN_Y_CHUNKED / KTF_TDNN_LO_PREFIX of KTF_GEMM_F16X2) */

#define KTF_TDNN_MX_LOADER (1 << 24)  /* ktf_tdnn_mx / ktf_tdnn_mx_stats only: the loader-wave kernel (csrc/tdnn_mxl.hip: 192 x 256 tiles,
                                   * eight matrix waves + four loader waves). It reads the weight images of its own
                                   * (mx.weight_images_loader; layout below) and, with KTF_TDNN_DET_STATS, writes one slot per 96-row
                                   * block: `sums` is (B, ktf_mx_stats_slots(T, flags), 2, units), reduced with slot_rows =
                                   * ktf_mx_slot_rows(flags) */
/* (1 << 25 was KTF_TDNN_MX_SLAB, the slab form of the 256-row kernel: measured as fast as the gathering kernel, not faster; 1 << 26 was
 * KTF_TDNN_MX_PERSIST, its persistent form -- one workgroup per CU walking its tiles, the operand ring running on across tile boundaries, the
 * previous tile encoded from registers under the next tile's first K-step: correct, 6 % slower. Both live on under tools/mx/experiments/.) */

/* name of the kernel family the calling thread's last ktf_tdnn* / ktf_tdnn_mx* call launched ("" before the first; a static
 * string). For the dispatch tests: which kernel a (gemm mode, layer shape) pair runs on is part of the library's contract. */
const char* ktf_tdnn_last_kernel(void);


/* number of output rows for an utterance with `len` input rows (tdnn.py:224-234) */
int64_t ktf_tdnn_out_len(int64_t len, const KtfTdnnDesc* d);
/* ... for a batch's lengths on the device: out_lens[b] = ktf_tdnn_out_len(lens[b], d). (ktf_tdnn writes them itself; the entry points on
 * the MX planes have no such argument: a VALID-padded or subsampling layer there is followed by this call.) */
int ktf_tdnn_out_lens(const int32_t* lens, int64_t B, const KtfTdnnDesc* d, int32_t* out_lens, void* stream);

int ktf_tdnn(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
             const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift,
             void* y, int64_t ldy, int32_t* out_lens, void* stream);

/* ktf_tdnn fused with the reducing StatsPooling that follows it (sequential.py:68-79 order "tdnn5 -> stats"): the layer
 * output is never written; instead sums[b, 0, u] += sum_t y[b,t,u] and sums[b, 1, u] += sum_t y[b,t,u]^2 (fp64, over
 * the valid rows). `sums` (B, 2, units) must be zeroed by the caller before the call (the order of the fp64 atomic adds
 * of an utterance's row blocks is not fixed: results can differ in the last fp64 bits from run to run; see
 * KTF_TDNN_DET_STATS for the reproducible form). Implemented by the 16-bit ring kernels -- KTF_GEMM_BF16 (bf16 x),
 * KTF_GEMM_BF16X3 (fp32 x, w_lo given): units > 128, SAME padding, subsampling 1 -- and by the bf16-pair small
 * tiles (KTF_GEMM_BF16X4: any layer shape; its KTF_TDNN_DET_STATS slots are ktf_tdnn_stats_slots(T, gemm) of
 * ktf_tdnn_slot_rows(gemm) = 64 rows). */
int ktf_tdnn_stats(const void* x, int64_t B, int64_t T, int64_t ldx, const int32_t* lens, const KtfTdnnDesc* d,
                   const void* w, const void* w_lo, const float* bias, const float* scale, const float* shift, double* sums,
                   void* stream);
/* Split-bf16 activations kept as TWO bf16 planes (hi = bf16(v), lo = bf16(v - hi)) instead of fp32: same bytes, but the
 * GEMM stages them as they are and its K-loop carries no fp32 -> (hi, lo) conversion (28 % shorter on MI355X). desc->gemm
 * must be KTF_GEMM_BF16X3 with x_dtype KTF_BF16; units > 128. y_dtype KTF_BF16 with y_lo != NULL writes the output as
 * planes again (the next layer's input); y_dtype KTF_F32 (y_lo NULL) writes plain fp32. ktf_split_bf16 produces the planes
 * of the first layer's input from fp32 rows (columns [D, ld_dst) are zeroed). */
int ktf_tdnn_split(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* lens,
                   const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                   const float* shift, void* y, void* y_lo, int64_t ldy, int32_t* out_lens, void* stream);
/* ktf_tdnn_split with the M-tiles laid over the batch's VALID rows end to end instead of 256-row tiles per utterance (short
 * utterances: a 1.5 s window is 148 rows, 0.58 of a tile). row_starts: (B + 1) int32 on the device, the exclusive prefix sums of the
 * utterance lengths (row_starts[0] = 0, row_starts[B] = the number of valid rows); a row's context offsets clamp against its own
 * utterance as in ktf_tdnn_split, and rows at or beyond an utterance's length are not written. KTF_GEMM_BF16X3 on row-major hi / lo
 * planes, SAME padding, no subsampling, ReLU or no activation, y as for ktf_tdnn_split (planes or fp32); B <= 4095 and
 * B * T * ldx * 2 < 2^32. Results equal ktf_tdnn_split's bit for bit (same operands into the same MFMAs in the same order). */
int ktf_tdnn_split_flat(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* row_starts, const int32_t* row_map,
                        const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                        const float* shift, void* y, void* y_lo, int64_t ldy, void* stream);
/* ktf_tdnn_split_flat fused with the reducing StatsPooling that follows it (ktf_tdnn_split_stats on the flat row tiles: the pooled
 * layer of 1.5 s windows otherwise computes 256-row tiles of 148 rows). `sums`: with KTF_TDNN_DET_STATS (B, ktf_flat_stats_slots(T), 2,
 * units) doubles, not zeroed by the caller: an utterance of len rows that starts at flat row s = row_starts[b] gets one partial sum per
 * 128-row block OF THE FLAT ROW SPACE it touches, in slots 0 .. ((s + len - 1) >> 7) - (s >> 7), and ktf_stats_finalize_flat adds
 * exactly those in slot order: reproducible run to run; the partition of an utterance's rows into partial sums depends on where the
 * batch places it, so its pooled values can differ in the last fp64 bits from batch to batch. Without the flag (B, 2, units), zeroed by
 * the caller, fp64 atomics (finalize with ktf_stats_finalize). */
int64_t ktf_flat_stats_slots(int64_t T);
/* The row table of the flat tiles: 4 int32 per flat row R < ktf_flat_row_map_rows(B, T) = round_up(B * T, 256) -- (output row b * T + t, or
 * -1 at and beyond row_starts[B]; frame t; length of the row's utterance; b). `row_map` of ktf_tdnn_split_flat / _flat_stats: NULL (every
 * workgroup then derives its 256 entries from row_starts: four dependent loads in front of its first DMA, ~3 us per tile) or this table,
 * made once per batch for all its layers -- same rows, same results. */
int64_t ktf_flat_row_map_rows(int64_t B, int64_t T);
int ktf_flat_row_map(const int32_t* row_starts, int64_t B, int64_t T, int32_t* map, void* stream);
int ktf_tdnn_split_flat_stats(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* row_starts,
                              const int32_t* row_map, const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                              const float* shift, double* sums, void* stream);
int ktf_stats_finalize_flat(const double* sums, int64_t slots, const int32_t* row_starts, int64_t T, int64_t B, int32_t D,
                            int32_t include_std, float eps, float* out, int64_t ld_out, void* stream);
int ktf_tdnn_split_stats(const void* x_hi, const void* x_lo, int64_t B, int64_t T, int64_t ldx, const int32_t* lens,
                         const KtfTdnnDesc* d, const void* w, const void* w_lo, const float* bias, const float* scale,
                         const float* shift, double* sums, void* stream);
int ktf_split_bf16(const float* src, int64_t rows, int32_t D, int64_t ld_src, void* hi, void* lo, int64_t ld_dst,
                   void* stream);
/* ... of a ragged batch: src (B, T, ld_src) -> planes (B, T, ld_dst); only the rows t < lens[b] are converted (lens NULL: all T): the
 * consumers clamp their row reads to the utterance, so the rest is never read. Same values as ktf_split_bf16 on the rows it writes. */
int ktf_split_bf16_rows(const float* src, int64_t B, int64_t T, int32_t D, int64_t ld_src, const int32_t* lens, void* hi, void* lo,
                        int64_t ld_dst, void* stream);
/* KTF_GEMM_F16MX (tdnn.py:251-280 + ReLU + BatchNorm, as ktf_tdnn). Activations are FOUR chunk-major planes; with nch = ceil(D / 32)
 * and record r = (b * nch + d / 32) * T + t of element (b, t, d):
 *   xh  : r * 64 B  32 halves, x_h = half(x) (saturating at +-65504), element d % 32
 *   xl4 : r * 16 B  32 e2m1 codes of x - x_h (element e in nibble e: byte e / 2, low nibble first)
 *   x4  : r * 16 B  32 e2m1 codes of x_h
 *   xs  : r * 4 B   uint32: byte 0 = E8M0 scale of the xl4 block, byte 1 = of the x4 block (value = code * 2^(scale - 127));
 *                   scale = floor(log2(block max)) - 2, one more when the maximum would round past 6
 * ktf_mx_planes makes them from fp32 rows (B, T, ld_src) (rows >= lens[b] are left unwritten); the layer itself writes them for
 * the next layer (yh, yl4, y4, ys; D = units), or fp32 rows (yf, ldy), or -- ktf_tdnn_mx_stats -- the pooled sums of
 * ktf_tdnn_stats (same `sums` layouts, KTF_TDNN_DET_STATS honoured). Exactly one output form per call; the others NULL.
 * Weights, for N-tile nt (256 units, zero rows beyond `units`) and K-step ks of the KTF_TDNN_K_INTERLEAVED order, K-steps
 * zero-padded to a multiple of 4 (nkp; a super-step ss = K-steps 4 ss .. 4 ss + 3):
 *   wh : block (nt * nkp + ks) * 16 KiB: the KTF_TDNN_W_TILED image of w_h
 *   wq : block (nt * nkp / 4 + ss) * 48 KiB, with kb = ks % 4, col = unit % 256, record q = kb * 256 + col:
 *          [0, 16 Ki)       q * 16: 32 e2m1 codes of w (K-step ks, unit col)
 *          [16 Ki, 32 Ki)   q * 16: bits 0..127 of the 32 e2m3 codes (6 bits each, element e at bit 6 e) of w - w_h
 *          [32 Ki, 40 Ki)   q * 8 : bits 128..191 of the same
 *          [40 Ki, 44 Ki)   q * 4 : uint32, byte 0 = E8M0 scale of the e2m1 block, byte 1 = of the e2m3 block
 *          [44 Ki, 48 Ki)   unused
 * ReLU or no activation; scale / shift as in ktf_tdnn (NULL when the BatchNorm is folded into the next layer,
 * TDNN.device_weights_mx). VALID padding and subsampling (tdnn.py:224-249) on the 256-row kernel: the output (planes or fp32 rows)
 * then has ktf_tdnn_out_len(T, d) rows per utterance and the valid rows of utterance b are ktf_tdnn_out_len(lens[b], d)
 * (ktf_tdnn_out_lens makes them for the next layer); ktf_tdnn_mx_stats and KTF_TDNN_MX_LOADER take SAME padding without
 * subsampling only. The 256-row kernel keeps a table of its K-steps in LDS: at most 1144 of them (ceil(D / 32) * nctx, padded to a multiple
 * of 4; K <= 36,608); the planes are read through 32-bit buffer offsets: T * din_pad * 2 < 2^31 - 2^20.
 * With KtfTdnnDesc.flags & KTF_TDNN_MX_LOADER the call runs the loader-wave kernel (csrc/tdnn_mxl.hip: 192 x 256 tiles over the flat row
 * space b * T + t, eight matrix + four loader waves) on the SAME activation planes and on weight images of its own
 * (mx.weight_images_loader): every operand fragment is 64 lanes x 16 (8, 4) consecutive bytes, and inside each 32-unit chunk the image
 * columns hold the units in the order (m >> 2) * 8 + (cb & 1) * 4 + (m & 3) for column m of unit block cb:
 *   wh : block (nt * nkp + ks) * 16 KiB: 16 unit-block fragments x 1 KiB, lane (q, m) = halves 8 q .. 8 q + 7 of the block's unit m
 *   wq : block (nt * nkp / 4 + ss) * 44 KiB = two halves of 22 KiB (unit blocks with (cb >> 1) & 1 == h, in the order
 *        2 (cb >> 2) + (cb & 1)): e2m1 codes 8 x 1 KiB | e2m3 bits 0..127 8 x 1 KiB | e2m3 bits 128..191 8 x 512 B | scale words
 *        8 x 256 B, lane (kb, m) = K block kb of the super-step, unit m of the block.
 * It needs B * T < 2^31 and B * T * ceil(D / 32) < 2^32, writes planes only without scale / shift, and its KTF_TDNN_DET_STATS slots are
 * 96 rows each (ktf_mx_stats_slots / ktf_mx_slot_rows). */
int ktf_mx_planes(const float* src, int64_t B, int64_t T, int32_t D, int64_t ld_src, const int32_t* lens, void* xh, void* xl4,
                  void* x4, void* xs, void* stream);
int ktf_tdnn_mx(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* lens,
                const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias, const float* scale, const float* shift,
                void* yh, void* yl4, void* y4, void* ys, float* yf, int64_t ldy, void* stream);
/* ktf_tdnn_mx with a plane output on FLAT ROW TILES: the 256-row M-tiles cover the batch's valid rows laid end to end (row_starts, row_map:
 * ktf_flat_row_map; as ktf_tdnn_split_flat) instead of ceil(T / 256) tiles per utterance -- 998-frame utterances fill 3.9 of their 4 tiles, a
 * batch the VAD left ragged fewer. SAME padding, no subsampling; B <= 4095, B * T * din_pad * 2 < 2^32. The planes equal ktf_tdnn_mx's bit for
 * bit (same operands into the same MFMAs in the same order); rows at and beyond an utterance's length are not written. */
int ktf_tdnn_mx_flat(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* row_starts,
                     const int32_t* row_map, const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias, const float* scale,
                     const float* shift, void* yh, void* yl4, void* y4, void* ys, void* stream);
/* ... and ktf_tdnn_mx_stats on flat row tiles: `sums` as for ktf_tdnn_split_flat_stats ((B, ktf_flat_stats_slots(T), 2, units) with
 * KTF_TDNN_DET_STATS, reduced by ktf_stats_finalize_flat; else (B, 2, units), zeroed by the caller) */
int ktf_tdnn_mx_flat_stats(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* row_starts,
                           const int32_t* row_map, const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias, const float* scale,
                           const float* shift, double* sums, void* stream);
int ktf_tdnn_mx_stats(const void* xh, const void* xl4, const void* x4, const void* xs, int64_t B, int64_t T, const int32_t* lens,
                      const KtfTdnnDesc* d, const void* wh, const void* wq, const float* bias, const float* scale,
                      const float* shift, double* sums, void* stream);
/* finishes the fused pooling: out[b, c] = sums[b,0,c]/n_b, out[b, D+c] = sqrt(max(sums[b,1,c]/n_b - mean^2, 0) + eps)
 * with n_b = lens[b] (or T when lens is NULL); stats_pooling.py:231-240. */
int ktf_stats_finalize(const double* sums, const int32_t* lens, int64_t T, int64_t B, int32_t D, int32_t include_std,
                       float eps, float* out, int64_t ld_out, void* stream);
/* KTF_TDNN_DET_STATS layout: number of 128-row slots of an utterance of T rows (2 * ceil(T / 256): whole 256-row tiles),
 * and the finalize that adds the slots 0 .. ceil(n_b / slot_rows) - 1 of sums (B, slots, 2, D) in that order before forming
 * mean / std as above (slot_rows = 128 for ktf_tdnn_stats / ktf_tdnn_split_stats, ktf_mx_slot_rows(flags) for ktf_tdnn_mx_stats). */
int64_t ktf_stats_slots(int64_t T);
/* ... of ktf_tdnn_stats for a GEMM mode: KTF_GEMM_BF16X4 (x, w of KTF_BF16P; the small-tile kernel) writes one slot per 64-row tile
 * (ceil(T / 64) slots of ktf_tdnn_slot_rows(gemm) = 64 rows), every other mode ktf_stats_slots(T) slots of 128 rows */
int64_t ktf_tdnn_stats_slots(int64_t T, int32_t gemm);
int32_t ktf_tdnn_slot_rows(int32_t gemm);
/* ... for ktf_tdnn_mx_stats, whose slot geometry depends on the kernel (KtfTdnnDesc.flags & KTF_TDNN_MX_LOADER: 96-row slots, two per
 * 192-row tile; else as ktf_stats_slots) */
int64_t ktf_mx_stats_slots(int64_t T, int32_t flags);
int32_t ktf_mx_slot_rows(int32_t flags);
int ktf_stats_finalize_slots(const double* sums, int64_t slots, int32_t slot_rows, const int32_t* lens, int64_t T, int64_t B, int32_t D,
                             int32_t include_std, float eps, float* out, int64_t ld_out, void* stream);

/* in place over the rows t < lens[b] of y (B, T, ld) fp32 (lens NULL: all T rows): y = act(y) * scale + shift per column, any KTF_ACT_*
 * (KTF_ACT_SOFTMAX: over the D units of each row, then the affine); scale / shift may be NULL (both or neither). The pass ktf_tdnn
 * appends for activations its epilogues do not fuse; rows at and beyond lens[b] are not touched. */
int ktf_activation_f32(float* y, int64_t B, int64_t T, int32_t D, int64_t ld, const int32_t* lens, int32_t act, const float* scale,
                       const float* shift, void* stream);

/* elementwise y = act(x) * scale + shift per column (stand-alone ReLU / BatchNorm layers; any KTF_ACT_* but KTF_ACT_SOFTMAX);
 * scale/shift may be NULL */
int ktf_affine_act_f32(const float* x, int64_t rows, int32_t D, int32_t act, const float* scale, const float* shift,
                       float* y, void* stream);
/* dtype conversion / column padding: dst (rows, ld_dst) <- src (rows, D) with zero fill of the pad columns */
int ktf_convert_pad(const void* src, int32_t src_dtype, int64_t rows, int32_t D, int64_t ld_src, void* dst,
                    int32_t dst_dtype, int64_t ld_dst, void* stream);

/* ------------------------------------------------------------------ statistics pooling (a11)
 * StatsPooling.computeStatsAcrossAll  layers/stats/stats_pooling.py:211-240
 * out (B, ld_out) fp32, columns [0,D) = mean_t and [D,2D) = sqrt(max(E[x^2]-mean^2,0)+eps) over rows
 * 0, input_period, ... < lens[b]; columns beyond are left untouched. */
int ktf_stats_pool(const void* x, int32_t x_dtype, int64_t B, int64_t T, int32_t D, int64_t ldx,
                   const int32_t* lens, int32_t input_period, int32_t include_std, float eps, float* out,
                   int64_t ld_out, void* stream);
/* StatsPooling.computeStatsAcrossWindows  layers/stats/stats_pooling.py:179-209,242-295 (fp32).
 * Output row j covers input rows start + j*output_period + {left..min(right, T-1) step input_period} inside [0,T). */
int ktf_stats_pool_windowed_f32(const float* x, int64_t B, int64_t T, int32_t D, int32_t left, int32_t right,
                                int32_t input_period, int32_t output_period, int32_t start, int64_t T_out,
                                int32_t include_std, float eps, float* out, void* stream);

/* The tail of the extractor in ONE launch: pooled statistics -> the affine after the pooling (tdnn6; sequential.py:68-79, W
 * (units, ldw) fp32 row-major, rows 16-byte aligned, ldw a multiple of 4 >= in_dim = (include_std ? 2 : 1) * D) -> x - mean -> LDA
 * A (units, out_dim) + off -> length normalisation (xvector_extractor.py:174-184). Input: fp32 pooled rows (`pooled`, ld_pooled;
 * stats_pooling.py:231-240 already applied) OR the fp64 sums of ktf_tdnn_stats / ktf_tdnn_mx_stats (`sums`, `slots`, `slot_rows` as for
 * ktf_stats_finalize[_slots]; the finalize happens in the kernel). Workspace: `partial` (B, 64, out_dim) fp32, `counters` (B)
 * uint32 ZEROED once by the caller (the kernel leaves them zero). `h_out` (B, units), optional: the affine's output.
 * Grid = 64 unit slices x ceil(B / group) utterance groups: a workgroup keeps its slice of W in registers and walks `group`
 * utterances (1 for a single utterance: 64 CUs share the 6 MB of W; ~32 for a large batch: W is read once per group). units <=
 * 512, in_dim <= 3072. Sums in a fixed order that does not depend on B or group: reproducible, batch == single bit for bit. */
int ktf_xvec_tail_f32(const float* pooled, int64_t ld_pooled, const double* sums, int64_t slots, int32_t slot_rows, const int32_t* lens, int64_t T,
                      int64_t B, int32_t D, int32_t include_std, float eps, const float* W, int64_t ldw, const float* bias,
                      int32_t units, const float* mean, const float* A, const float* off, int32_t out_dim, float* partial,
                      uint32_t* counters, float* y, float* h_out, int32_t group, int32_t flags, void* stream);
#define KTF_TAIL_SKIP_EMPTY 1     /* ktf_xvec_tail_f32 flags: utterances with lens[b] == 0 are skipped and their rows of y left as they are
                                   * (the second, tighter pass over the short utterances of a batch: XvectorExtractor.route_short_utterances) */
/* ------------------------------------------------------------------ x-vector post-processing (a12)
 * models/kaldi/xvector_extractor.py:174-184: y = (x - mean) @ A + off ; y *= sqrt(out)/||y||_2
 * x (B, in) fp32, A (in, out) row-major, off (out). */
int ktf_xvec_post_f32(const float* x, int64_t B, int32_t in_dim, int32_t out_dim, const float* mean, const float* A,
                      const float* off, float* y, void* stream);

/* ------------------------------------------------------------------ PLDA (a16)
 * PLDA.call  layers/plda/plda.py:247-263. fp64 (reference default) and fp32 variants.
 * x (B, dim); A (dim, dim) row-major; offset = -A*mean (dim); psi (dim).
 * transformed (B, dim); scores (B, B) with scores[i, j] = LLR(x_i | class of x_j). */
int ktf_plda_f64(const double* x, int64_t B, int32_t dim, const double* A, const double* offset, const double* psi,
                 int32_t normalize_length, int32_t simple_length_norm, double* transformed, double* scores,
                 void* stream);
int ktf_plda_f32(const float* x, int64_t B, int32_t dim, const float* A, const float* offset, const float* psi,
                 int32_t normalize_length, int32_t simple_length_norm, float* transformed, float* scores,
                 void* stream);

/* Rectangular trial blocks (plda.py:198-245 logLikelihoodRatio on two sets; the reference only scores a batch against
 * itself): scores[i, j] (N x M, row-major) = LLR(test_tr[i] | class of enroll_tr[j]); both inputs are TRANSFORMED vectors
 * (ktf_plda_* with scores == NULL produces them). Lets a large trial matrix be cut into row blocks, one per GPU. */
int ktf_plda_score_f64(const double* test_tr, int64_t N, const double* enroll_tr, int64_t M, int32_t dim,
                       const double* psi, double* scores, void* stream);
int ktf_plda_score_f32(const float* test_tr, int64_t N, const float* enroll_tr, int64_t M, int32_t dim, const float* psi,
                       float* scores, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KTF_HIP_H_ */
