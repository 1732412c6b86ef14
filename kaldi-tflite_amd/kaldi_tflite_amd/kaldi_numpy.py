"""
ktf.kaldi_numpy — caller-side NumPy helpers with the reference's names and semantics
(kaldi_tflite/lib/kaldi_numpy/frame_extraction.py). `PadWaveform` is the step a caller applies
before `Framing` to get Kaldi's snip-edges=false framing; the others are host-side reference
implementations users of the original package rely on in their own tests.
"""

import numpy as np

from .ops import window_function


def MirrorPad(x, left_pad, right_pad):
    """frame_extraction.py:28-51."""
    left = np.flip(x[..., :left_pad], axis=-1)
    right = np.flip(x[..., -right_pad:], axis=-1)
    return np.concatenate([left, x, right], axis=-1)


def PadWaveform(x, frameSize, frameShift):
    """frame_extraction.py:54-89: mirror-pad so that Framing yields round(N/shift) frames."""
    N = x.shape[-1]
    M = (N + frameShift // 2) // frameShift
    Nv = (M - 1) * frameShift + frameSize
    leftOver = abs(N - Nv)
    leftPad = (frameSize - frameShift) // 2
    return MirrorPad(x, leftPad, leftOver - leftPad)


def ExtractFrames(samples, frameSizeMs, frameShiftMs, sampleFreq, snipEdges):
    """frame_extraction.py:92-188 (strided view of the frames)."""
    m = int(sampleFreq * frameSizeMs / 1000.0)
    k = int(sampleFreq * frameShiftMs / 1000.0)
    N = samples.shape[-1]
    if snipEdges:
        M = 1 + (N - m) // k
        N = (M - 1) * k + m
    x = samples[:N]
    shape = x.shape[:-1] + (N - m + 1, m)
    strides = x.strides + (x.strides[-1],)
    return np.lib.stride_tricks.as_strided(x, shape=shape, strides=strides)[::k]


def ProcessFrames(frames, dither=0.0, remove_dc_offset=True, preemphasis_coefficient=0.97, window_type="povey",
                  raw_energy=True):
    """frame_extraction.py:191-265 -> (windows, log energy)."""
    if preemphasis_coefficient < 0 or preemphasis_coefficient > 1:
        raise ValueError("preemphasis coefficient must be between 0 and 1")
    if window_type not in ("hanning", "hamming", "rectangular", "blackman", "povey", "sine"):
        raise ValueError(f"invalid window type {window_type}")
    M = frames.shape[-1]
    if M == 0:
        raise ValueError("window_size must be > 0")
    w = window_function(window_type, M).reshape([1] * (frames.ndim - 1) + [-1])
    eps = np.finfo(frames.dtype).eps
    win = frames.copy()
    if dither != 0.0:
        win += (np.random.normal(size=win.shape) * dither).astype(win.dtype)
    if remove_dc_offset:
        win = win - np.mean(win, axis=-1, keepdims=True)
    if raw_energy:
        energy = np.sum(np.power(win, 2), axis=-1, keepdims=True).clip(min=eps)
    if preemphasis_coefficient > 0.0:
        win[..., 1:] -= preemphasis_coefficient * win[..., :-1]
        win[..., 0] -= preemphasis_coefficient * win[..., 0]
    win = win * w
    if not raw_energy:
        energy = np.sum(np.power(win, 2), axis=-1, keepdims=True).clip(min=eps)
    return win, np.log(energy)


def GetWindowFunction(window_type, window_size):
    """frame_extraction.py:141-187: hamming | hanning | rectangular | blackman | povey | sine; ValueError for a zero
    size or an unknown name."""
    if window_size == 0:
        raise ValueError("window_size must be > 0")
    if window_type not in ("hamming", "hanning", "rectangular", "blackman", "povey", "sine"):
        raise ValueError(f"invalid window type {window_type}")
    return window_function(window_type, window_size)


def getWindowedSums(frames, N, padding):
    """frame_extraction.py:268-322: sums over sliding windows of N frames along axis -2 of a zero-prefixed array; with
    "SAME" padding the edge outputs repeat the first / last full window."""
    return _windowed_sums(frames, N, padding)


def _windowed_sums(frames, N, padding):
    T = frames.shape[-2] - 1
    cs = np.cumsum(frames, axis=-2)
    s = cs[..., N:, :] - cs[..., :-N, :]
    if padding == "VALID":
        return s
    a, b = N // 2, T - (N - 1) // 2
    out = np.zeros(frames.shape[:-2] + (T, frames.shape[-1]), dtype=frames.dtype)
    out[..., a:b, :] = s
    out[..., :a, :] = out[..., a:a + 1, :]
    out[..., b:, :] = out[..., b - 1:b, :]
    return out


def ApplyCMVN(frames, center=False, norm_vars=False, window=600, min_window=100, padding="SAME"):
    """frame_extraction.py:325-400."""
    if not center:
        raise NotImplementedError("ApplyCMVN with center=False not supported yet")
    padding = padding.upper()
    if padding not in ["SAME", "VALID"]:
        raise ValueError(f"`padding` should be either 'SAME' or 'VALID', got '{padding}'")
    T = frames.shape[-2]
    N = window
    std = 1
    if T <= N:
        mean = np.mean(frames, axis=-2, keepdims=True)
        if norm_vars:
            std = np.std(frames, axis=-2, keepdims=True)
        return np.divide(frames - mean, std)
    pad = [[0, 0] for _ in range(frames.ndim)]
    pad[-2] = [1, 0]
    padded = np.pad(frames, pad, mode="constant")
    mean = _windowed_sums(padded, N, padding) / N
    if norm_vars:
        std = np.sqrt(_windowed_sums(np.power(padded, 2), N, padding) / N - np.power(mean, 2))
    if padding == "VALID":
        a, b = N // 2, T - (N - 1) // 2
        return np.divide(frames[..., a:b, :] - mean, std)
    return np.divide(frames - mean, std)
