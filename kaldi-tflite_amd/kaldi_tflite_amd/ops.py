"""
Thin torch-tensor wrappers over the C-ABI (one function per entry point of include/ktf_hip.h)
plus the host-side constant tables the kernels consume (window, FFT twiddles, sparse mel bank,
DCT matrix, lifter), computed once in float64 exactly as the reference's layer `build()`
methods do and uploaded as fp32.

PyTorch is used for device memory and streams only; all arithmetic on activations happens in
the HIP kernels.
"""

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _dev(device=None):
    L.require_gpu()
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def default_device():
    return _dev(None)


def to_device_f32(x, device=None):
    """numpy / tensor -> contiguous fp32 tensor on the GPU."""
    dev = _dev(device)
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)), device=dev)


def round_up(n, m):
    return (n + m - 1) // m * m


# ----------------------------------------------------------------------------- host-side tables
def window_function(window_type, M, blackman_coeff=0.42):
    """layers/dsp/windowing.py:110-156 of the reference (float64)."""
    t = window_type.lower()
    n = np.arange(0, M)
    if M == 1:
        return np.ones(1, float)
    if t == "hamming":
        return np.hamming(M)
    if t == "hanning":
        return np.hanning(M)
    if t == "povey":
        return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / (M - 1))) ** 0.85
    if t == "rectangular":
        return np.ones((M,))
    if t == "sine":
        return np.sin(np.pi * n / (M - 1))
    if t == "blackman":
        w = np.blackman(M)
        if blackman_coeff != 0.42:
            w = w - 0.42 + blackman_coeff
        return w
    raise ValueError(f"window_type '{window_type}' is not recognized")


def next_power_of_2(n):
    if n & (n - 1) == 0 and n != 0:
        return n
    return 2 ** (n - 1).bit_length()


def mel_bank_dense(window_size, num_bins, sample_freq, lower, upper):
    """layers/dsp/filterbank.py:141-189: dense (nfft/2+1, num_bins) fp32 bank (weights in fp64, strict
    left < mel < right, no weight on the Nyquist bin)."""
    nfft = next_power_of_2(window_size)
    bins = nfft // 2
    bw = sample_freq / nfft
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)  # noqa: E731
    mlo, mhi = mel(lower), mel(upper)
    delta = (mhi - mlo) / (num_bins + 1)
    bank = np.zeros([num_bins, bins + 1], dtype=np.float32)
    m = mel(bw * np.arange(bins))
    for i in range(num_bins):
        left = mlo + i * delta
        center = left + delta
        right = center + delta
        inside = (m > left) & (m < right)
        up = (m - left) / (center - left)
        down = (right - m) / (right - center)
        bank[i, :bins] = np.where(inside, np.where(m <= center, up, down), 0.0)
    return nfft, bank.T.copy()


def dct_matrix(input_length, length):
    """layers/dsp/dct.py:98-143 (float64, (input_length, length); column 0 overwritten with sqrt(1/N))."""
    N = float(input_length)
    n = np.arange(input_length)
    k = np.arange(length, dtype=np.float64)[:, None]
    d = np.cos(np.pi / N * (n + 0.5) * k)
    d[0] *= 1.0 / np.sqrt(2.0)
    d *= np.sqrt(2.0 / N)
    d = d.T
    d[:, 0] = np.sqrt(1.0 / N)
    return d


def lifter_coeffs(num_mfccs, q):
    n = np.arange(0, num_mfccs)
    return 1 + 0.5 * np.sin(np.pi * n / q) * q


def fast512_tables(starts, lens, w, maxw=16):
    """Per-lane constants of frontend512.hip: FFT twiddles of the lane-level radix-4 schedule and the split of the
    sparse mel bank into <= 64 (filter, <= maxw-bin slice) work items with <= 4 adjacent lanes per filter."""
    lane = np.arange(64)
    n0 = (lane >> 1) + 32 * (lane & 1)
    n1 = (lane >> 1) & 15
    n2 = (lane >> 1) & 3
    tw = np.zeros((64, 18), np.float64)
    for r in (1, 2, 3):
        for base, nn, N in ((0, n0, 256), (6, n1, 64), (12, n2, 16)):
            ang = -2.0 * np.pi * nn * r / N
            tw[:, base + 2 * (r - 1)] = np.cos(ang)
            tw[:, base + 2 * (r - 1) + 1] = np.sin(ang)
    def split(width):
        its = []                                # (filter, start, len)
        for f, (s0, ln) in enumerate(zip(starts, lens)):
            if ln == 0:
                its.append((f, s0, 0))
                continue
            parts = -(-ln // width)
            if parts > 4:
                return None
            step = -(-ln // parts)
            for p in range(parts):
                a = p * step
                b = min(ln, a + step)
                its.append((f, s0 + a, b - a))
        return its if len(its) <= 64 else None

    items, used = None, None
    for width in range(8, maxw + 1):            # smallest slice width that fits 64 lanes with <= 4 lanes per filter
        items = split(width)
        if items is not None:
            used = max(ln for _, _, ln in items)
            break
    if items is None:
        return None
    meta = np.zeros((64, 4), np.int32)
    meta[:, 2] = -1
    mw = np.zeros((64, maxw), np.float32)
    for i, (f, s0, ln) in enumerate(items):
        meta[i, 0], meta[i, 1], meta[i, 2] = s0, ln, f
        mw[i, :ln] = w[f, s0 - starts[f]: s0 - starts[f] + ln]
    for i, (f, _, _) in enumerate(items):
        fl = 0
        if i + 1 < len(items) and items[i + 1][0] == f:
            fl |= 1
        if i + 2 < len(items) and items[i + 2][0] == f:
            fl |= 2
        if i == 0 or items[i - 1][0] != f:
            fl |= 4
        meta[i, 3] = fl
    return tw, meta, mw, max(int(used), 1)


class FrontendTables:
    """Device copies of every constant ktf_frontend_f32 needs, for one (frame_size, mel, dct) configuration."""

    def __init__(self, frame_size, window=None, mel_bank=None, dct=None, lifter=None, device=None):
        dev = _dev(device)
        self.frame_size = int(frame_size)
        self.nfft = next_power_of_2(self.frame_size)
        n2 = self.nfft // 2
        k = np.arange(n2)
        tw = np.stack([np.cos(2 * np.pi * k / n2), -np.sin(2 * np.pi * k / n2)], -1)
        rtw = np.stack([np.cos(2 * np.pi * k / self.nfft), -np.sin(2 * np.pi * k / self.nfft)], -1)
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)  # noqa: E731
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=dev)  # noqa: E731
        self.window = f32(window) if window is not None else None
        self.twiddle = f32(tw.reshape(-1))
        self.rtwiddle = f32(rtw.reshape(-1))
        self.mel_start = self.mel_len = self.mel_w = None
        self.mel_stride = 0
        self.num_mels = 0
        if mel_bank is not None:
            bank = np.asarray(mel_bank, dtype=np.float32)          # (nfft/2+1, num_mels)
            assert bank.shape[0] == n2 + 1
            self.num_mels = bank.shape[1]
            starts, lens = [], []
            for f in range(self.num_mels):
                nz = np.nonzero(bank[:n2, f])[0]
                if nz.size == 0:
                    starts.append(0)
                    lens.append(0)
                else:
                    starts.append(int(nz[0]))
                    lens.append(int(nz[-1] - nz[0] + 1))
            self.mel_stride = max(1, max(lens))
            w = np.zeros((self.num_mels, self.mel_stride), np.float32)
            for f in range(self.num_mels):
                w[f, :lens[f]] = bank[starts[f]:starts[f] + lens[f], f]
            self.mel_start, self.mel_len, self.mel_w = i32(starts), i32(lens), f32(w)
        self.dct = f32(dct) if dct is not None else None
        self.lifter = f32(lifter) if lifter is not None else None
        # tables of the register-resident nfft = 512 fast path (frontend512.hip)
        self.fast_tw = self.fast_mel_meta = self.fast_mel_w = None
        fast_maxw = 0
        if self.nfft == 512 and mel_bank is not None and self.num_mels <= 32:
            fast = fast512_tables(starts, lens, w)
            if fast is not None:
                self.fast_tw, self.fast_mel_meta, self.fast_mel_w = f32(fast[0]), i32(fast[1]), f32(fast[2])
                fast_maxw = fast[3]
        self.struct = L.FrontendTables(
            window=L.ptr(self.window), twiddle=L.ptr(self.twiddle), rtwiddle=L.ptr(self.rtwiddle),
            mel_start=L.ptr(self.mel_start), mel_len=L.ptr(self.mel_len), mel_w=L.ptr(self.mel_w),
            dct=L.ptr(self.dct), lifter=L.ptr(self.lifter), fast_tw=L.ptr(self.fast_tw),
            fast_mel_meta=L.ptr(self.fast_mel_meta), fast_mel_w=L.ptr(self.fast_mel_w), mel_stride=self.mel_stride,
            reserved=fast_maxw)


# ----------------------------------------------------------------------------- entry-point wrappers
def num_frames(n_samples, frame_size, frame_shift):
    return int(L.load().ktf_num_frames(int(n_samples), int(frame_size), int(frame_shift)))


def frontend(x, in_kind, cfg, tables, out_stage, n, B, T, seed=0, want_energy=False, out=None):
    """x: fp32 device tensor (int16 for L.IN_WAV_I16). Returns out (B,T,last) [, energy (B,T)]."""
    lib = L.load()
    last = {L.OUT_FRAMES: cfg.frame_size, L.OUT_WINDOWED: cfg.frame_size, L.OUT_FBANK: cfg.num_mels,
            L.OUT_MFCC: cfg.num_ceps}[out_stage]
    if out is None:
        out = torch.empty((B, T, last), dtype=torch.float32, device=x.device)
    energy = torch.empty((B, T), dtype=torch.float32, device=x.device) if want_energy else None
    with L.on_device(x.device):
        rc = lib.ktf_frontend_f32(L.ptr(x), B, n, in_kind, C.byref(cfg), C.byref(tables.struct), out_stage, L.ptr(out),
                                  L.ptr(energy), C.c_uint64(seed & (2**64 - 1)), L.stream_ptr())
    L.check(rc, "ktf_frontend_f32")
    return (out, energy) if want_energy else out


def dct(x2d, dct_t, lifter_t, out_dim):
    lib = L.load()
    rows, in_dim = x2d.shape
    out = torch.empty((rows, out_dim), dtype=torch.float32, device=x2d.device)
    with L.on_device(x2d.device):
        rc = lib.ktf_dct_f32(L.ptr(x2d), rows, in_dim, out_dim, L.ptr(dct_t), L.ptr(lifter_t), L.ptr(out), L.stream_ptr())
    L.check(rc, "ktf_dct_f32")
    return out


def vad_mask(feats, cfg):
    lib = L.load()
    B, T, D = feats.shape
    mask = torch.empty((B, T), dtype=torch.float32, device=feats.device)
    with L.on_device(feats.device):
        rc = lib.ktf_vad_mask_f32(L.ptr(feats), B, T, D, C.byref(cfg), L.ptr(mask), L.stream_ptr())
    L.check(rc, "ktf_vad_mask_f32")
    return mask


def vad_index(feats, cfg):
    lib = L.load()
    B, T, D = feats.shape
    idx = torch.empty((B, T), dtype=torch.int32, device=feats.device)
    lens = torch.empty((B,), dtype=torch.int32, device=feats.device)
    with L.on_device(feats.device):
        rc = lib.ktf_vad_index(L.ptr(feats), B, T, D, C.byref(cfg), L.ptr(idx), L.ptr(lens), L.stream_ptr())
    L.check(rc, "ktf_vad_index")
    return idx, lens


def cmvn(x, cfg, lens=None, ldo=None, want_lens=False):
    lib = L.load()
    B, T, D = x.shape
    ldo = D if ldo is None else ldo
    out = torch.empty((B, T, ldo), dtype=torch.float32, device=x.device)
    work = torch.empty((B * T * 2 * D + 2 * D,), dtype=torch.float32, device=x.device)
    out_lens = torch.empty((B,), dtype=torch.int32, device=x.device) if want_lens else None
    with L.on_device(x.device):
        rc = lib.ktf_cmvn_f32(L.ptr(x), B, T, D, x.stride(1), L.ptr(lens), C.byref(cfg), L.ptr(out), ldo, L.ptr(out_lens),
                              L.ptr(work), L.stream_ptr())
    L.check(rc, "ktf_cmvn_f32")
    return (out, out_lens) if want_lens else out


def vad_cmvn(feats, vad_cfg, cmvn_cfg, out, lens, idx_work, work):
    lib = L.load()
    B, T, D = feats.shape
    dt = L.ktf_dtype(out.dtype)
    with L.on_device(feats.device):
        rc = lib.ktf_vad_cmvn(L.ptr(feats), B, T, D, C.byref(vad_cfg), C.byref(cmvn_cfg), L.ptr(out), dt, out.stride(1),
                              L.ptr(lens), L.ptr(idx_work), L.ptr(work), L.stream_ptr())
    L.check(rc, "ktf_vad_cmvn")


def last_kernel():
    """Kernel family the calling thread's last ktf_tdnn* / ktf_tdnn_mx* call launched (include/ktf_hip.h)."""
    return (L.load().ktf_tdnn_last_kernel() or b"").decode()


def build_id():
    """16 hex digits naming the sources the loaded library was built from (ktf_build_id)."""
    return (L.load().ktf_build_id() or b"").decode()


def clock_probe(out, us, stream):
    """Launches the one-wave shader-clock probe on `stream` (a torch.cuda.Stream of its own, next to the measured work): out is a
    (4,) int64 CUDA tensor that receives shader clocks, 100 MHz ticks, lowest / highest kHz over 1 ms windows (ktf_clock_probe)."""
    with L.on_device(out.device):
        rc = L.load().ktf_clock_probe(L.ptr(out), int(us), stream.cuda_stream)
    L.check(rc, "ktf_clock_probe")
    return out


def route_short(lens, min_frames, lens_main, lens_short, host_flag=None, seq=0):
    """lens -> (lens_main, lens_short) by voiced length (ktf_route_short); host_flag: pinned int32[2] CPU tensor that receives the number
    of short utterances and then `seq`."""
    lib = L.load()
    with L.on_device(lens.device):
        rc = lib.ktf_route_short(L.ptr(lens), lens.shape[0], int(min_frames), L.ptr(lens_main), L.ptr(lens_short),
                                 host_flag.data_ptr() if host_flag is not None else None, int(seq), L.stream_ptr())
    L.check(rc, "ktf_route_short")


def tdnn_out_len(T, desc):
    return int(L.load().ktf_tdnn_out_len(int(T), C.byref(desc)))


def tdnn_out_lens(lens, desc, out):
    """out[b] = tdnn_out_len(lens[b]) on the device (for the layers behind ktf_tdnn_mx, which has no out_lens argument)."""
    with L.on_device(lens.device):
        rc = L.load().ktf_tdnn_out_lens(L.ptr(lens), lens.shape[0], C.byref(desc), L.ptr(out), L.stream_ptr())
    L.check(rc, "ktf_tdnn_out_lens")
    return out


def tdnn(x, lens, desc, w, w_lo, bias, scale, shift, y, out_lens=None):
    """x (B,T,ldx) fp32/bf16, y (B,Tout,ldy) preallocated."""
    lib = L.load()
    B, T = x.shape[0], x.shape[1]
    with L.on_device(x.device):
        rc = lib.ktf_tdnn(L.ptr(x), B, T, x.stride(1), L.ptr(lens), C.byref(desc), L.ptr(w), L.ptr(w_lo), L.ptr(bias),
                          L.ptr(scale), L.ptr(shift), L.ptr(y), y.stride(1), L.ptr(out_lens), L.stream_ptr())
    L.check(rc, "ktf_tdnn")
    return y


def stats_slots(T, mx_flags=None):
    """Slots of the reproducible pooling layout (KTF_TDNN_DET_STATS) for utterances of up to T rows: 128-row slots, or -- for
    ktf_tdnn_mx_stats, `mx_flags` = its KtfTdnnDesc.flags -- the slots of the MX kernel in use (mx_slot_rows rows each)."""
    if mx_flags is None:
        return int(L.load().ktf_stats_slots(int(T)))
    return int(L.load().ktf_mx_stats_slots(int(T), int(mx_flags)))


def tdnn_stats_slots(T, gemm):
    """Slots / rows per slot of ktf_tdnn_stats with KTF_TDNN_DET_STATS for a GEMM mode (KTF_GEMM_BF16X4: one per 64-row tile)."""
    return int(L.load().ktf_tdnn_stats_slots(int(T), int(gemm)))


def tdnn_slot_rows(gemm):
    return int(L.load().ktf_tdnn_slot_rows(int(gemm)))


def mx_slot_rows(mx_flags):
    return int(L.load().ktf_mx_slot_rows(int(mx_flags)))


def tdnn_stats(x, lens, desc, w, w_lo, bias, scale, shift, sums, zero=True):
    """Fused TDNN + reducing stats pooling: accumulates fp64 column sums / sums of squares into `sums` (B,2,units), or
    stores them per 128-row block into (B,slots,2,units) when desc.flags has TDNN_DET_STATS (zero=False then)."""
    lib = L.load()
    B, T = x.shape[0], x.shape[1]
    with L.on_device(x.device):
        if zero:
            sums.zero_()
        rc = lib.ktf_tdnn_stats(L.ptr(x), B, T, x.stride(1), L.ptr(lens), C.byref(desc), L.ptr(w), L.ptr(w_lo), L.ptr(bias),
                                L.ptr(scale), L.ptr(shift), L.ptr(sums), L.stream_ptr())
    L.check(rc, "ktf_tdnn_stats")
    return sums


def plda_score(test_tr, enroll_tr, psi):
    """scores (N, M) of transformed test vectors against transformed enrollment vectors."""
    lib = L.load()
    N, dim = test_tr.shape
    M = enroll_tr.shape[0]
    scores = torch.empty((N, M), dtype=test_tr.dtype, device=test_tr.device)
    fn = lib.ktf_plda_score_f64 if test_tr.dtype == torch.float64 else lib.ktf_plda_score_f32
    with L.on_device(test_tr.device):
        rc = fn(L.ptr(test_tr), N, L.ptr(enroll_tr), M, dim, L.ptr(psi), L.ptr(scores), L.stream_ptr())
    L.check(rc, "ktf_plda_score")
    return scores


def split_bf16(src, D, planes, lens=None):
    """fp32 (B,T,ld_src) rows -> planes (2,B,T,ld) bf16: hi = bf16(v), lo = bf16(v - hi); pad columns zero. `lens`: only the rows
    t < lens[b] are converted (the consumers never read the rest)."""
    lib = L.load()
    with L.on_device(src.device):
        if lens is None:
            rc = lib.ktf_split_bf16(L.ptr(src), src.shape[0] * src.shape[1], D, src.stride(1), L.ptr(planes[0]), L.ptr(planes[1]), planes.shape[-1],
                                    L.stream_ptr())
        else:
            rc = lib.ktf_split_bf16_rows(L.ptr(src), src.shape[0], src.shape[1], D, src.stride(1), L.ptr(lens), L.ptr(planes[0]), L.ptr(planes[1]),
                                         planes.shape[-1], L.stream_ptr())
    L.check(rc, "ktf_split_bf16")
    return planes


def _planes(xp):
    """(hi, lo, B, T, ldx) of a (2,B,T,ldx) pair of bf16 planes."""
    if xp.dim() != 4 or xp.shape[0] != 2:
        raise ValueError(f"expected a (2, B, T, ldx) pair of bf16 planes, got shape {tuple(xp.shape)}")
    return xp[0], xp[1], xp.shape[1], xp.shape[2], xp.stride(2)


def tdnn_split(xp, lens, desc, w, w_lo, bias, scale, shift, y, y_lo=None, out_lens=None):
    """xp: (2,B,T,ldx) bf16 hi/lo planes. y: (B,Tout,ldy) bf16 plane (+ y_lo) or fp32."""
    lib = L.load()
    hi, lo, B, T, ldx = _planes(xp)
    with L.on_device(xp.device):
        rc = lib.ktf_tdnn_split(L.ptr(hi), L.ptr(lo), B, T, ldx, L.ptr(lens), C.byref(desc), L.ptr(w), L.ptr(w_lo),
                                L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(y), L.ptr(y_lo), y.stride(1), L.ptr(out_lens),
                                L.stream_ptr())
    L.check(rc, "ktf_tdnn_split")
    return y


def row_starts(lens, B, T, out):
    """out (B + 1,) int32, out[0] == 0 -> the exclusive prefix sums of the utterance lengths (all T when lens is None):
    ktf_tdnn_split_flat's row map."""
    if lens is None:
        out[1:] = torch.arange(1, B + 1, dtype=torch.int32, device=out.device) * int(T)
    else:
        torch.cumsum(lens, 0, dtype=torch.int32, out=out[1:])
    return out


class FlatRows:
    """Row bookkeeping of a batch on flat row tiles: `starts` (B + 1,) int32 prefix sums of the lengths, `map` the per-row table
    ktf_flat_row_map makes from them (or None: every workgroup derives its entries itself)."""

    def __init__(self, starts, map=None):
        self.starts, self.map = starts, map


def flat_rows(lens, B, T, get):
    """FlatRows of a batch: `get(role, shape, dtype)` hands out the two buffers (a workspace)."""
    lib = L.load()
    starts = row_starts(lens, B, T, get("row_starts", (B + 1,), torch.int32))
    table = get("row_map", (int(lib.ktf_flat_row_map_rows(B, T)), 4), torch.int32)
    with L.on_device(starts.device):
        rc = lib.ktf_flat_row_map(L.ptr(starts), B, T, L.ptr(table), L.stream_ptr())
    L.check(rc, "ktf_flat_row_map")
    return FlatRows(starts, table)


def _flat(starts):
    return (starts.starts, starts.map) if isinstance(starts, FlatRows) else (starts, None)


def tdnn_split_flat(xp, starts, desc, w, w_lo, bias, scale, shift, y, y_lo=None):
    """tdnn_split with M-tiles over the batch's valid rows laid end to end (`starts` = row_starts(lens, ...) or flat_rows(...)): short
    utterances."""
    lib = L.load()
    hi, lo, B, T, ldx = _planes(xp)
    starts, rmap = _flat(starts)
    with L.on_device(xp.device):
        rc = lib.ktf_tdnn_split_flat(L.ptr(hi), L.ptr(lo), B, T, ldx, L.ptr(starts), L.ptr(rmap), C.byref(desc), L.ptr(w), L.ptr(w_lo),
                                     L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(y), L.ptr(y_lo), y.stride(1), L.stream_ptr())
    L.check(rc, "ktf_tdnn_split_flat")
    return y


def tdnn_split_stats(xp, lens, desc, w, w_lo, bias, scale, shift, sums, zero=True):
    lib = L.load()
    hi, lo, B, T, ldx = _planes(xp)
    with L.on_device(xp.device):
        if zero:
            sums.zero_()
        rc = lib.ktf_tdnn_split_stats(L.ptr(hi), L.ptr(lo), B, T, ldx, L.ptr(lens), C.byref(desc), L.ptr(w),
                                      L.ptr(w_lo), L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(sums), L.stream_ptr())
    L.check(rc, "ktf_tdnn_split_stats")
    return sums


def flat_stats_slots(T):
    return int(L.load().ktf_flat_stats_slots(int(T)))


def tdnn_split_flat_stats(xp, starts, desc, w, w_lo, bias, scale, shift, sums, zero=False):
    """tdnn_split_stats on flat row tiles (`starts` = row_starts(lens, ...)). sums: (B, flat_stats_slots(T), 2, units) fp64 with
    KTF_TDNN_DET_STATS in desc.flags (not zeroed: stats_finalize_flat reads the slots that were written), else (B, 2, units), zero=True."""
    lib = L.load()
    hi, lo, B, T, ldx = _planes(xp)
    starts, rmap = _flat(starts)
    with L.on_device(xp.device):
        if zero:
            sums.zero_()
        rc = lib.ktf_tdnn_split_flat_stats(L.ptr(hi), L.ptr(lo), B, T, ldx, L.ptr(starts), L.ptr(rmap), C.byref(desc), L.ptr(w), L.ptr(w_lo),
                                           L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(sums), L.stream_ptr())
    L.check(rc, "ktf_tdnn_split_flat_stats")
    return sums


def stats_finalize_flat(sums, starts, T, D, include_std, eps, out, slots):
    """sums (B, slots, 2, D) fp64 of tdnn_split_flat_stats -> out (B, ld) mean | std."""
    lib = L.load()
    B = sums.shape[0]
    starts, _ = _flat(starts)
    with L.on_device(sums.device):
        rc = lib.ktf_stats_finalize_flat(L.ptr(sums), slots, L.ptr(starts), T, B, D, int(include_std), eps, L.ptr(out), out.stride(0),
                                         L.stream_ptr())
    L.check(rc, "ktf_stats_finalize_flat")
    return out


def mx_planes(src, D, lens, planes):
    """fp32 (B,T,ld) rows -> the four KTF_GEMM_F16MX planes (mx.Planes); rows >= lens[b] are left unwritten."""
    lib = L.load()
    B, T = src.shape[0], src.shape[1]
    with L.on_device(src.device):
        rc = lib.ktf_mx_planes(L.ptr(src), B, T, D, src.stride(1), L.ptr(lens), L.ptr(planes.xh), L.ptr(planes.xl4), L.ptr(planes.x4),
                               L.ptr(planes.xs), L.stream_ptr())
    L.check(rc, "ktf_mx_planes")
    return planes


def tdnn_mx(xp, lens, desc, wh, wq, bias, scale, shift, y):
    """xp: mx.Planes. y: mx.Planes (the next layer's input) or an fp32 tensor, (B, tdnn_out_len(T), ...) either way."""
    lib = L.load()
    B, T, _ = xp.shape
    planes = not isinstance(y, torch.Tensor)
    with L.on_device(xp.device):
        rc = lib.ktf_tdnn_mx(L.ptr(xp.xh), L.ptr(xp.xl4), L.ptr(xp.x4), L.ptr(xp.xs), B, T, L.ptr(lens), C.byref(desc), L.ptr(wh),
                             L.ptr(wq), L.ptr(bias), L.ptr(scale), L.ptr(shift),
                             L.ptr(y.xh) if planes else None, L.ptr(y.xl4) if planes else None, L.ptr(y.x4) if planes else None,
                             L.ptr(y.xs) if planes else None, None if planes else L.ptr(y), 0 if planes else y.stride(1), L.stream_ptr())
    L.check(rc, "ktf_tdnn_mx")
    return y


def tdnn_mx_flat(xp, rows, desc, wh, wq, bias, scale, shift, y):
    """tdnn_mx with a plane output on flat row tiles (`rows` = flat_rows(lens, ...)): the same planes, bit for bit."""
    lib = L.load()
    B, T, _ = xp.shape
    with L.on_device(xp.device):
        rc = lib.ktf_tdnn_mx_flat(L.ptr(xp.xh), L.ptr(xp.xl4), L.ptr(xp.x4), L.ptr(xp.xs), B, T, L.ptr(rows.starts), L.ptr(rows.map), C.byref(desc),
                                  L.ptr(wh), L.ptr(wq), L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(y.xh), L.ptr(y.xl4), L.ptr(y.x4),
                                  L.ptr(y.xs), L.stream_ptr())
    L.check(rc, "ktf_tdnn_mx_flat")
    return y


def tdnn_mx_flat_stats(xp, rows, desc, wh, wq, bias, scale, shift, sums, zero=False):
    """tdnn_mx_stats on flat row tiles; sums as for tdnn_split_flat_stats (finalize: stats_finalize_flat)."""
    lib = L.load()
    B, T, _ = xp.shape
    with L.on_device(xp.device):
        if zero:
            sums.zero_()
        rc = lib.ktf_tdnn_mx_flat_stats(L.ptr(xp.xh), L.ptr(xp.xl4), L.ptr(xp.x4), L.ptr(xp.xs), B, T, L.ptr(rows.starts), L.ptr(rows.map),
                                        C.byref(desc), L.ptr(wh), L.ptr(wq), L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(sums), L.stream_ptr())
    L.check(rc, "ktf_tdnn_mx_flat_stats")
    return sums


def tdnn_mx_stats(xp, lens, desc, wh, wq, bias, scale, shift, sums, zero=True):
    lib = L.load()
    B, T, _ = xp.shape
    with L.on_device(xp.device):
        if zero:
            sums.zero_()
        rc = lib.ktf_tdnn_mx_stats(L.ptr(xp.xh), L.ptr(xp.xl4), L.ptr(xp.x4), L.ptr(xp.xs), B, T, L.ptr(lens), C.byref(desc),
                                   L.ptr(wh), L.ptr(wq), L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(sums), L.stream_ptr())
    L.check(rc, "ktf_tdnn_mx_stats")
    return sums


def stats_finalize(sums, lens, T, D, include_std, eps, out, slots=0, slot_rows=128):
    """sums (B,2,D) [slots == 0] or (B,slots,2,D) fp64 (one slot per `slot_rows` rows) -> out (B, ld) mean | std."""
    lib = L.load()
    B = sums.shape[0]
    with L.on_device(sums.device):
        if slots:
            rc = lib.ktf_stats_finalize_slots(L.ptr(sums), slots, int(slot_rows), L.ptr(lens), T, B, D, int(include_std), eps, L.ptr(out),
                                              out.stride(0), L.stream_ptr())
        else:
            rc = lib.ktf_stats_finalize(L.ptr(sums), L.ptr(lens), T, B, D, int(include_std), eps, L.ptr(out), out.stride(0),
                                        L.stream_ptr())
    L.check(rc, "ktf_stats_finalize")
    return out


def affine_act(x, act, scale=None, shift=None):
    lib = L.load()
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    with L.on_device(x.device):
        rc = lib.ktf_affine_act_f32(L.ptr(x), rows, D, act, L.ptr(scale), L.ptr(shift), L.ptr(y), L.stream_ptr())
    L.check(rc, "ktf_affine_act_f32")
    return y


def activation_(y, lens, act, scale=None, shift=None):
    """In place over the rows t < lens[b] of y (B, T, D) fp32: y = act(y) * scale + shift (any L.ACT_*, softmax over D)."""
    lib = L.load()
    B, T, D = y.shape
    assert y.dtype == torch.float32 and y.stride(2) == 1 and y.stride(0) == T * y.stride(1)
    with L.on_device(y.device):
        rc = lib.ktf_activation_f32(L.ptr(y), B, T, D, y.stride(1), L.ptr(lens), act, L.ptr(scale), L.ptr(shift), L.stream_ptr())
    L.check(rc, "ktf_activation_f32")
    return y


def pair_encode(t):
    """fp32 tensor -> the KTF_BF16P pairs of its values, as a float32 tensor of the same shape (raw bits: bits 0-15 = bf16(v), round
    to nearest even, bits 16-31 = bf16(v - bf16(v)))."""
    t = t.to(torch.float32).contiguous()
    hi = t.to(torch.bfloat16)
    lo = (t - hi.to(torch.float32)).to(torch.bfloat16)
    bits = (hi.view(torch.int16).to(torch.int32) & 0xFFFF) | (lo.view(torch.int16).to(torch.int32) << 16)
    return bits.view(torch.float32)


def pair_decode(t):
    """The values a KTF_BF16P tensor holds (hi + lo), fp32."""
    bits = t.contiguous().view(torch.int32)
    hi = (bits << 16).view(torch.float32)
    lo = (bits & ~0xFFFF).view(torch.float32)
    return hi + lo


def convert_pad(src, D, dst):
    """src (..., ld_src) / dst (..., ld_dst) 2-D-viewable row-major tensors; copies D columns, zero-fills the pad."""
    lib = L.load()
    rows = src.numel() // src.shape[-1]
    sd, dd = L.ktf_dtype(src.dtype), L.ktf_dtype(dst.dtype)
    with L.on_device(src.device):
        rc = lib.ktf_convert_pad(L.ptr(src), sd, rows, D, src.shape[-1], L.ptr(dst), dd, dst.shape[-1], L.stream_ptr())
    L.check(rc, "ktf_convert_pad")
    return dst


def stats_pool(x, D, lens, input_period, include_std, eps, out):
    """x (B,T,ldx) fp32/bf16; out (B, ld_out) fp32 preallocated."""
    lib = L.load()
    B, T = x.shape[0], x.shape[1]
    dt = L.ktf_dtype(x.dtype)
    with L.on_device(x.device):
        rc = lib.ktf_stats_pool(L.ptr(x), dt, B, T, D, x.stride(1), L.ptr(lens), input_period, int(include_std), eps,
                                L.ptr(out), out.stride(0), L.stream_ptr())
    L.check(rc, "ktf_stats_pool")
    return out


def stats_pool_windowed(x, left, right, input_period, output_period, start, T_out, include_std, eps):
    lib = L.load()
    B, T, D = x.shape
    out = torch.empty((B, T_out, 2 * D if include_std else D), dtype=torch.float32, device=x.device)
    with L.on_device(x.device):
        rc = lib.ktf_stats_pool_windowed_f32(L.ptr(x), B, T, D, left, right, input_period, output_period, start, T_out,
                                             int(include_std), eps, L.ptr(out), L.stream_ptr())
    L.check(rc, "ktf_stats_pool_windowed_f32")
    return out


def xvec_post(x, mean, A, off, out=None):
    lib = L.load()
    B, in_dim = x.shape
    out_dim = A.shape[1]
    if out is None:
        out = torch.empty((B, out_dim), dtype=torch.float32, device=x.device)
    with L.on_device(x.device):
        rc = lib.ktf_xvec_post_f32(L.ptr(x), B, in_dim, out_dim, L.ptr(mean), L.ptr(A), L.ptr(off), L.ptr(out), L.stream_ptr())
    L.check(rc, "ktf_xvec_post_f32")
    return out


def xvec_tail(pooled, sums, slots, lens, T, D, include_std, eps, W, bias, units, mean, A, off, partial, counters, out, h_out=None, group=1,
              slot_rows=128, skip_empty=False):
    """Fused tail (ktf_xvec_tail_f32): pooled (B, ld) fp32 rows OR fp64 sums -> tdnn6 -> mean-sub -> LDA -> length norm, one launch."""
    lib = L.load()
    B = out.shape[0]
    src = pooled if pooled is not None else sums
    with L.on_device(src.device):
        rc = lib.ktf_xvec_tail_f32(L.ptr(pooled), pooled.stride(0) if pooled is not None else 0, L.ptr(sums), int(slots), int(slot_rows), L.ptr(lens), int(T), B,
                                   int(D), int(include_std), float(eps), L.ptr(W), W.stride(0), L.ptr(bias), int(units), L.ptr(mean), L.ptr(A),
                                   L.ptr(off), A.shape[1], L.ptr(partial), L.ptr(counters), L.ptr(out), L.ptr(h_out), int(group),
                                   L.TAIL_SKIP_EMPTY if skip_empty else 0, L.stream_ptr())
    L.check(rc, "ktf_xvec_tail_f32")
    return out


def plda(x, A, offset, psi, normalize_length, simple_length_norm, want_scores=True):
    lib = L.load()
    B, dim = x.shape
    tr = torch.empty_like(x)
    scores = torch.empty((B, B), dtype=x.dtype, device=x.device) if want_scores else None
    fn = lib.ktf_plda_f64 if x.dtype == torch.float64 else lib.ktf_plda_f32
    with L.on_device(x.device):
        rc = fn(L.ptr(x), B, dim, L.ptr(A), L.ptr(offset), L.ptr(psi), int(normalize_length), int(simple_length_norm),
                L.ptr(tr), L.ptr(scores), L.stream_ptr())
    L.check(rc, "ktf_plda")
    return scores, tr
