"""
ktf.kaldi_numpy — host-side NumPy helpers under the reference's names
(kaldi_tflite/lib/kaldi_numpy/frame_extraction.py): what a caller of the original package uses
around the layers (mirror padding for Kaldi's snip-edges=false framing) and in its own tests
(frames, windowed frames + log energy, sliding-window CMVN).

On the device the same padding is `Framing(snip_edges=False)` (KtfFrontendCfg.pad_mode = 1, fused
into the frame gather), so nothing here is on the hot path.
"""

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

from .ops import window_function

_WINDOWS = ("hamming", "hanning", "rectangular", "blackman", "povey", "sine")


def MirrorPad(x, left_pad, right_pad):
    """Reflect the signal about its two ends, edge sample included (… x1 x0 | x0 x1 … xn | xn xn-1 …);
    frame_extraction.py:28-51."""
    x = np.asarray(x)
    left_pad, right_pad = int(left_pad), int(right_pad)
    if 0 <= left_pad <= x.shape[-1] and 0 < right_pad <= x.shape[-1]:
        return np.pad(x, [(0, 0)] * (x.ndim - 1) + [(left_pad, right_pad)], mode="symmetric")
    # degenerate requests (shift > size makes the head negative; a zero tail): the reference's slice arithmetic is what
    # its callers see, so keep it -- x[:l] and x[-r:] reversed, whatever those slices select
    return np.concatenate([x[..., :left_pad][..., ::-1], x, x[..., -right_pad:][..., ::-1]], axis=-1)


def PadWaveform(x, frameSize, frameShift):
    """Pads so that snip-edges framing of the result gives Kaldi's snip-edges=false frame count,
    round(N / shift), with the first frame centred half a shift into the signal (frame_extraction.py:54-89)."""
    n = x.shape[-1]
    frames = (n + frameShift // 2) // frameShift
    span = (frames - 1) * frameShift + frameSize          # samples those frames cover
    head = (frameSize - frameShift) // 2
    return MirrorPad(x, head, abs(n - span) - head)


def ExtractFrames(samples, frameSizeMs, frameShiftMs, sampleFreq, snipEdges):
    """(…, N) -> (…, frames, size) view: every `shift`-th window of `size` samples that fits
    (frame_extraction.py:92-138; `snipEdges` only trims the tail first, which selects the same windows)."""
    size, shift = int(sampleFreq * frameSizeMs / 1000.0), int(sampleFreq * frameShiftMs / 1000.0)
    return sliding_window_view(samples, size, axis=-1)[..., ::shift, :]


def GetWindowFunction(window_type, window_size):
    """frame_extraction.py:141-187."""
    if window_size == 0:
        raise ValueError("window_size must be > 0")
    if window_type not in _WINDOWS:
        raise ValueError(f"invalid window type {window_type}")
    return window_function(window_type, window_size)


def ProcessFrames(frames, dither=0.0, remove_dc_offset=True, preemphasis_coefficient=0.97, window_type="povey",
                  raw_energy=True):
    """Kaldi's per-frame chain dither -> DC removal -> [log energy] -> pre-emphasis -> window -> [log energy]
    (frame_extraction.py:191-265). Returns (windowed frames, log energy (…, 1)); the energy is floored at the dtype's eps."""
    if not 0.0 <= preemphasis_coefficient <= 1.0:
        raise ValueError("preemphasis coefficient must be between 0 and 1")
    window = GetWindowFunction(window_type, frames.shape[-1])
    floor = np.finfo(frames.dtype).eps

    def log_energy(v):
        return np.log(np.maximum(np.sum(v * v, axis=-1, keepdims=True), floor))

    y = np.array(frames)
    if dither != 0.0:
        y += (dither * np.random.normal(size=y.shape)).astype(y.dtype)
    if remove_dc_offset:
        y = y - y.mean(axis=-1, keepdims=True)
    energy = log_energy(y) if raw_energy else None
    if preemphasis_coefficient > 0.0:
        prev = np.concatenate([y[..., :1], y[..., :-1]], axis=-1)      # the first sample is its own predecessor
        y = y - preemphasis_coefficient * prev
    y = y * window
    return y, (energy if raw_energy else log_energy(y))


def getWindowedSums(frames, N, padding):
    """`frames` is zero-prefixed along axis -2 (T + 1 rows). "VALID": the T - N + 1 sums of N consecutive rows;
    "SAME": T rows, row t = the window starting at clip(t - N//2, 0, T - N) (frame_extraction.py:268-322)."""
    T = frames.shape[-2] - 1
    running = np.cumsum(frames, axis=-2)
    first = np.arange(T - N + 1)
    if padding != "VALID":
        first = np.clip(np.arange(T) - N // 2, 0, T - N)
    return np.take(running, first + N, axis=-2) - np.take(running, first, axis=-2)


def ApplyCMVN(frames, center=False, norm_vars=False, window=600, min_window=100, padding="SAME"):
    """Sliding-window mean (and variance) normalisation over axis -2, window centred on the frame and shifted inside the
    utterance at its ends; an utterance no longer than the window is normalised globally (frame_extraction.py:325-400)."""
    if not center:
        raise NotImplementedError("ApplyCMVN with center=False not supported yet")
    padding = padding.upper()
    if padding not in ("SAME", "VALID"):
        raise ValueError(f"`padding` should be either 'SAME' or 'VALID', got '{padding}'")
    T = frames.shape[-2]
    if T <= window:
        scale = np.std(frames, axis=-2, keepdims=True) if norm_vars else 1
        return (frames - np.mean(frames, axis=-2, keepdims=True)) / scale
    lead = [(0, 0)] * frames.ndim
    lead[-2] = (1, 0)
    z = np.pad(frames, lead)
    mean = getWindowedSums(z, window, padding) / window
    scale = np.sqrt(getWindowedSums(z * z, window, padding) / window - mean * mean) if norm_vars else 1
    if padding == "VALID":
        frames = frames[..., window // 2: T - (window - 1) // 2, :]
    return (frames - mean) / scale
