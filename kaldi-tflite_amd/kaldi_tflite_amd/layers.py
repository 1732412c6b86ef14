"""
ktf.layers — the reference's operator API (Keras layer protocol) on torch (ROCm) tensors.

Same class names, constructor kwargs, defaults, return conventions and exceptions as
kaldi_tflite/lib/layers/* of the reference; each `call` launches hand-written HIP kernels
through the C-ABI (see include/ktf_hip.h) instead of TensorFlow ops. The one deliberate
API difference is the tensor type (torch.Tensor on an MI355X instead of tf.Tensor).
"""

import collections
import itertools
import zlib

import numpy as np
import torch

from . import _lib as L
from . import ops

_name_counters = collections.defaultdict(int)


def _auto_name(cls_name):
    base = "".join(("_" + c.lower()) if c.isupper() and i else c.lower() for i, c in enumerate(cls_name))
    base = {"m_f_c_c": "mfcc", "v_a_d": "vad", "c_m_v_n": "cmvn", "t_d_n_n": "tdnn", "d_c_t": "dct", "p_l_d_a": "plda",
            "re_l_u": "re_lu"}.get(base, base)
    n = _name_counters[base]
    _name_counters[base] += 1
    return base if n == 0 else f"{base}_{n}"


def _to_device(x):
    """numpy / CPU tensor -> device tensor (float64 is kept, everything else becomes float32)."""
    dev = ops._dev()
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            x = x.to(dev)
        if x.dtype not in (torch.float32, torch.float64, torch.bfloat16):
            x = x.to(torch.float32)
        return x
    a = np.asarray(x)
    dt = np.float64 if a.dtype == np.float64 else np.float32
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)


def _per_device(layer, device, make):
    """One device-resident constant set per (layer, device): a layer called on a second GPU must not hand kernels
    pointers into the first one's memory."""
    cache = layer.__dict__.setdefault("_dev_cache", {})
    key = str(device)
    if key not in cache:
        cache[key] = make()
    return cache[key]


class Layer:
    """Minimal stand-in for tf.keras.layers.Layer: name, build-on-first-call, get_config/from_config, weights."""

    def __init__(self, trainable=False, name=None, dtype="float32", **kwargs):
        if kwargs:
            # keras rejects unknown kwargs except a known few; mirror the permissive ones used by the reference tests
            allowed = {"input_shape", "batch_input_shape", "batch_size", "weights", "activity_regularizer", "autocast",
                       "implementation", "reduce"}
            bad = set(kwargs) - allowed
            if bad:
                raise TypeError(f"Keyword argument not understood: {sorted(bad)}")
        self.name = name if name is not None else _auto_name(type(self).__name__)
        self.trainable = trainable
        self.dtype = dtype
        self.built = False

    def build(self, input_shape):
        self.built = True

    def _maybe_build(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape))
            self.built = True

    def __call__(self, inputs, *args, **kwargs):
        inputs = _to_device(inputs)
        self._maybe_build(inputs)
        return self.call(inputs, *args, **kwargs)

    def call(self, inputs):
        raise NotImplementedError

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype}

    @classmethod
    def from_config(cls, config):
        # `trainable` and `dtype` are the base layer's own entries of get_config() (keras.layers.Layer takes them back); the layers here fix
        # both in their constructors, so a get_config() -> from_config() round trip drops them instead of passing them twice
        config = {k: v for k, v in dict(config).items() if k not in ("trainable", "dtype")}
        return cls(**config)

    def get_weights(self):
        return []

    def set_weights(self, weights):
        if len(weights) != 0:
            raise ValueError(f"layer {self.name} has no weights, got {len(weights)}")

    def compute_output_shape(self, input_shape):
        return input_shape


# =============================================================================== dsp
class Framing(Layer):
    """layers/dsp/framing.py:25 — frames of `frame_length_ms` every `frame_shift_ms`, no padding.

    Extensions (not in the reference): `snip_edges=False` frames the waveform as Kaldi's --snip-edges=false does, i.e. as
    if it had been mirror-padded with the reference's `kaldi_numpy.PadWaveform` (frame_extraction.py:54-89) first — done
    inside the frame gather on the device; int16 tensors are consumed as they are (no fp32 copy)."""

    def __init__(self, frame_length_ms=25.0, frame_shift_ms=10.0, sample_frequency=16000.0, name=None,
                 dynamic_input_shape=False, snip_edges=True, **kwargs):
        super().__init__(trainable=False, name=name, **kwargs)
        self.snipEdges = bool(snip_edges)
        self.sampleFreq = sample_frequency
        self.frameSizeMs = frame_length_ms
        self.frameShiftMs = frame_shift_ms
        self.dynamicInputShape = dynamic_input_shape
        self.batchAxis, self.sampleAxis = 0, -1
        if self.frameSizeMs <= 0 or self.frameShiftMs <= 0 or self.sampleFreq <= 0:
            raise ValueError("frame_length, frame_shift and sample_frequency should be > 0")
        self.frameSize = int(sample_frequency * frame_length_ms / 1000.0)
        self.frameShift = int(sample_frequency * frame_shift_ms / 1000.0)
        if self.frameSize <= 0:
            raise ValueError("frame_length should be high enough to contain at least 1 sample")
        if self.frameShift <= 0:
            raise ValueError("frame_shift should be high enough to shift by at least 1 sample")
        self.halfFrameSize = self.frameSize // 2
        self.frameWidth = 2 * self.halfFrameSize        # framing.py:104-106: offsets are range(-half, half)
        self.numInputSamples = None

    def build(self, input_shape):
        n = input_shape[self.sampleAxis]
        if n is None and not self.dynamicInputShape:
            raise ValueError("input_shape must not be unknown if dynamic_input_shape set to False")
        if n is not None:
            if n < self.minSamples():
                raise ValueError(f"input sample size (axis={self.sampleAxis}) must be >= frame size ({self.frameSize})")
            self.numInputSamples = n
        self.built = True

    def numFrames(self, n):
        if not self.snipEdges:
            # kaldi_numpy PadWaveform + Framing: M = round(n / shift) frames over the mirror-padded waveform
            M = (n + self.frameShift // 2) // self.frameShift
            Nv = (M - 1) * self.frameShift + self.frameWidth
            left = (self.frameWidth - self.frameShift) // 2
            right = (Nv - n) - left
            if M < 1 or Nv < n or left < 0 or right < 0 or left > n or right > n:
                raise ValueError(f"snip_edges=False: mirror padding is undefined for {n} samples")
            return M
        # centres = range(half, n - half + 1, shift)
        span = n - 2 * self.halfFrameSize
        return 0 if span < 0 else 1 + span // self.frameShift

    def minSamples(self):
        return self.frameSize if self.snipEdges else 1

    @staticmethod
    def device_samples(inputs):
        """(tensor, in_kind) the kernels read: int16 PCM stays int16, everything else becomes fp32."""
        x = inputs
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            if isinstance(x, np.ndarray) and x.dtype == np.int16:
                x = torch.as_tensor(x).to(ops.default_device())
            elif isinstance(x, torch.Tensor) and x.dtype == torch.int16:
                x = x.to(ops.default_device())
            else:
                x = ops.to_device_f32(x)
        kind = L.IN_WAV_I16 if x.dtype == torch.int16 else L.IN_WAV
        if x.dtype != torch.int16:
            x = x.to(torch.float32)
        # (B, N) views with unit sample stride are read in place: overlapping windows of one recording
        # (`wav.unfold(-1, N, hop)`, sliding-window / diarization extraction) are not materialised
        if not (x.dim() == 2 and x.stride(1) == 1 and 0 < x.stride(0) < 2**31):
            x = x.contiguous()
        return x, kind

    def compute_output_shape(self, input_shape):
        n = input_shape[self.sampleAxis]
        if n is None and not self.dynamicInputShape:
            raise ValueError("input_shape must not be unknown if dynamic_input_shape set to False")
        out = list(input_shape[:-1])
        out.extend([None if n is None else self.numFrames(n), self.frameWidth])
        return out

    def get_config(self):
        c = super().get_config()
        c.update({"frame_length": self.frameSizeMs, "frame_shift": self.frameShiftMs,
                  "sample_frequency": self.sampleFreq, "dynamic_input_shape": self.dynamicInputShape})
        if not self.snipEdges:
            c["snip_edges"] = False
        return c

    def _cfg(self):
        return L.FrontendCfg(frame_size=self.frameWidth, frame_shift=self.frameShift,
                             nfft=max(64, ops.next_power_of_2(self.frameWidth)), num_mels=1, num_ceps=1,
                             pad_mode=0 if self.snipEdges else 1)

    def call(self, inputs):
        x, kind = self.device_samples(inputs)
        n = x.shape[-1]
        if not self.dynamicInputShape and self.numInputSamples is not None and n != self.numInputSamples:
            raise ValueError(f"layer was built for {self.numInputSamples} samples, got {n}")
        if n < self.minSamples():
            raise ValueError(f"input sample size (axis={self.sampleAxis}) must be >= frame size ({self.frameSize})")
        lead = x.shape[:-1]
        B = int(np.prod(lead)) if len(lead) else 1
        T = self.numFrames(n)
        cfg = self._cfg()
        tables = _per_device(self, x.device, lambda: ops.FrontendTables(self.frameWidth, device=x.device))
        if x.dim() == 2 and not x.is_contiguous():
            cfg.row_stride = x.stride(0)
        else:
            x = x.reshape(B, n)
        out = ops.frontend(x, kind, cfg, tables, L.OUT_FRAMES, n, B, T)
        if self.dynamicInputShape:
            return out.reshape(lead[0] if len(lead) else 1, -1, self.frameWidth)
        return out.reshape(*lead, T, self.frameWidth)


class Windowing(Layer):
    """layers/dsp/windowing.py:27."""

    def __init__(self, window_type="povey", blackman_coeff=0.42, dither=0.0, remove_dc_offset=True,
                 preemphasis_coefficient=0.97, return_energy=True, raw_energy=True, energy_floor=0.0, epsilon=1e-7,
                 name=None, **kwargs):
        super().__init__(trainable=False, name=name, **kwargs)
        self.preemphasisCoeff = preemphasis_coefficient
        if self.preemphasisCoeff < 0 or self.preemphasisCoeff > 1.0:
            raise ValueError("preemphasis_coefficient should be between 0.0 and 1.0")
        self.windowType = window_type.lower()
        if self.windowType not in ["hamming", "hanning", "povey", "rectangular", "sine", "blackman"]:
            raise ValueError(f"window_type '{window_type}' is not recognized")
        self.blackmanCoeff = blackman_coeff
        self.dither = dither
        self.removeDCOffset = remove_dc_offset
        self.returnEnergy = return_energy
        self.rawEnergy = raw_energy
        self.energyFloor = energy_floor
        self.eps = float(epsilon)
        self.windowFunc = None
        self.sampleAxis = -1
        self._seed = itertools.count(0x5EED + 1)      # (next() of a count is atomic: concurrent callers never draw the same dither seed)

    def build(self, input_shape):
        M = input_shape[self.sampleAxis]
        if M == 0:
            raise ValueError(f"window size (input shape axis = {self.sampleAxis}) needs to be > 0")
        self.windowFunc = ops.window_function(self.windowType, M, self.blackmanCoeff)
        self._M = M
        self._dev_cache = {}
        self.built = True

    def get_config(self):
        c = super().get_config()
        c.update({"window_type": self.windowType, "blackman_coeff": self.blackmanCoeff, "dither": self.dither,
                  "remove_dc_offset": self.removeDCOffset, "preemphasis_coefficient": self.preemphasisCoeff,
                  "return_energy": self.returnEnergy, "raw_energy": self.rawEnergy, "energy_floor": self.energyFloor,
                  "epsilon": self.eps})
        return c

    def _cfg(self, M):
        return L.FrontendCfg(frame_size=M, frame_shift=M, nfft=max(64, ops.next_power_of_2(M)), num_mels=1, num_ceps=1,
                             remove_dc=int(self.removeDCOffset), raw_energy=int(self.rawEnergy),
                             use_energy=int(self.returnEnergy), preemph=self.preemphasisCoeff, dither=self.dither,
                             energy_floor=self.energyFloor, eps=self.eps)

    def call(self, inputs):
        x = inputs.to(torch.float32).contiguous()
        M = x.shape[-1]
        if M != self._M:
            raise ValueError(f"layer was built for frames of {self._M} samples, got {M}")
        lead = x.shape[:-1]
        rows = int(np.prod(lead))
        tables = _per_device(self, x.device, lambda: ops.FrontendTables(M, window=self.windowFunc, device=x.device))
        r = ops.frontend(x.reshape(1, rows, M), L.IN_FRAMES, self._cfg(M), tables, L.OUT_WINDOWED, rows, 1, rows,
                         seed=next(self._seed), want_energy=self.returnEnergy)
        if self.returnEnergy:
            out, en = r
            return out.reshape(*lead, M), en.reshape(*lead, 1)
        return r.reshape(*lead, M)


class FilterBank(Layer):
    """layers/dsp/filterbank.py:27 — zero-padded rFFT -> |.|(^2) -> mel bank -> log."""

    def __init__(self, num_bins=23, sample_frequency=16000.0, high_freq_cutoff=0.0, low_freq_cutoff=20.0,
                 use_log_fbank=True, use_power=True, epsilon=1e-7, name=None, **kwargs):
        super().__init__(trainable=False, name=name, **kwargs)
        self.numBins = num_bins
        if self.numBins <= 2:
            raise ValueError(f"num_bins must be >= 3, got {num_bins}")
        self.sampleFreq = sample_frequency
        self.nyquist = sample_frequency / 2.0
        if self.sampleFreq <= 0:
            raise ValueError(f"sample_frequency must be > 0, got {sample_frequency}")
        self.lowerCutoff = low_freq_cutoff
        if self.lowerCutoff > self.nyquist or self.lowerCutoff < 0:
            raise ValueError(f"low_freq_cutoff must be > 0 and < Nyquist Rate ({self.nyquist} Hz)")
        self.upperCutoff = high_freq_cutoff
        if self.upperCutoff <= 0:
            self.upperCutoff += self.nyquist
        if self.lowerCutoff >= self.upperCutoff:
            raise ValueError("lower_freq_cutoff must be < higher_freq_cutoff")
        self.useLogFBank = use_log_fbank
        self.usePower = use_power
        self.eps = float(epsilon)
        self.melBank = None
        self.fftLength = None
        self.sampleAxis = -1

    def build(self, input_shape):
        M = input_shape[self.sampleAxis]
        self.fftLength, self.melBank = ops.mel_bank_dense(M, self.numBins, self.sampleFreq, self.lowerCutoff, self.upperCutoff)
        self._M = M
        self._dev_cache = {}
        self.built = True

    def nextPowerOf2(self, n):
        return ops.next_power_of_2(n)

    def compute_output_shape(self, input_shape):
        out = list(input_shape)
        out[self.sampleAxis] = self.numBins
        return out

    def get_config(self):
        c = super().get_config()
        c.update({"sample_frequency": self.sampleFreq, "num_bins": self.numBins, "lower_freq_cutoff": self.lowerCutoff,
                  "upper_freq_cutoff": self.upperCutoff, "use_log_fbank": self.useLogFBank, "use_power": self.usePower,
                  "epsilon": self.eps})
        return c

    def call(self, inputs):
        x = inputs.to(torch.float32).contiguous()
        M = x.shape[-1]
        if M != self._M:
            raise ValueError(f"layer was built for frames of {self._M} samples, got {M}")
        lead = x.shape[:-1]
        rows = int(np.prod(lead))
        tables = _per_device(self, x.device, lambda: ops.FrontendTables(M, mel_bank=self.melBank, device=x.device))
        cfg = L.FrontendCfg(frame_size=M, frame_shift=M, nfft=max(64, self.fftLength), num_mels=self.numBins, num_ceps=1,
                            use_power=int(self.usePower), use_log=int(self.useLogFBank), eps=self.eps)
        out = ops.frontend(x.reshape(1, rows, M), L.IN_WINDOWED, cfg, tables, L.OUT_FBANK, rows, 1, rows)
        return out.reshape(*lead, self.numBins)


class DCT(Layer):
    """layers/dsp/dct.py:27 — orthonormal DCT-II as a matmul (column 0 = sqrt(1/N))."""

    def __init__(self, length, dct_type=2, norm="ortho", name=None, **kwargs):
        super().__init__(trainable=False, name=name, **kwargs)
        self.length = length
        if self.length <= 0:
            raise ValueError(f"DCT length must be > 0, got {length}")
        self.dctType = dct_type
        if self.dctType not in [2]:
            raise NotImplementedError(f"DCT-{dct_type} is not supported yet")
        self.norm = norm.lower()
        if self.norm not in ["ortho"]:
            raise NotImplementedError(f"{norm} normalization is not supported yet")
        self.dct = None
        self.featAxis = -1

    def build(self, input_shape):
        featDim = input_shape[self.featAxis]
        if featDim < self.length:
            raise ValueError("input feature length must be >= DCT length")
        self.dct = ops.dct_matrix(featDim, self.length)
        self._dev_cache = {}
        self.built = True

    def compute_output_shape(self, input_shape):
        batch, time, _ = input_shape
        return (batch, time, self.length)

    def get_config(self):
        c = super().get_config()
        c.update({"length": self.length, "dct_type": self.dctType, "norm": self.norm})
        return c

    def call(self, inputs):
        x = inputs.to(torch.float32).contiguous()
        dct_dev = _per_device(self, x.device, lambda: ops.to_device_f32(self.dct, x.device))
        lead = x.shape[:-1]
        out = ops.dct(x.reshape(-1, x.shape[-1]), dct_dev, None, self.length)
        return out.reshape(*lead, self.length)


class MFCC(Layer):
    """layers/dsp/mfcc.py:28 — Windowing -> FilterBank -> DCT -> lifter -> C0 <- log-energy, as ONE fused kernel."""

    def __init__(self, num_mfccs=23, num_mels=23, cepstral_lifter=22, use_energy=True, sample_frequency=16000.0,
                 high_freq_cutoff=0.0, low_freq_cutoff=20.0, use_log_fbank=True, use_power=True, window_type="povey",
                 dither=0.0, remove_dc_offset=True, preemphasis_coefficient=0.97, raw_energy=True, energy_floor=0.0,
                 epsilon=1e-7, name=None, **kwargs):
        super().__init__(trainable=False, name=name)
        self.numMfccs = num_mfccs
        self.melBins = num_mels
        self.cepstralLifter = cepstral_lifter
        self.useEnergy = use_energy
        if self.numMfccs > self.melBins:
            raise ValueError("num_mfccs must be <= num_mels")
        self.eps = float(epsilon)
        self.batchAxis, self.frameAxis, self.sampleAxis = 0, -2, -1
        self.lifters = ops.lifter_coeffs(self.numMfccs, self.cepstralLifter) if (self.numMfccs > 1 and self.cepstralLifter > 1) else None
        self.windowing = Windowing(window_type=window_type, dither=dither, remove_dc_offset=remove_dc_offset,
                                   preemphasis_coefficient=preemphasis_coefficient, raw_energy=raw_energy,
                                   return_energy=use_energy, energy_floor=energy_floor, epsilon=epsilon)
        self.filterbank = FilterBank(num_bins=num_mels, sample_frequency=sample_frequency,
                                     high_freq_cutoff=high_freq_cutoff, low_freq_cutoff=low_freq_cutoff,
                                     use_log_fbank=use_log_fbank, use_power=use_power, epsilon=epsilon)
        self.dct = DCT(length=num_mfccs, dct_type=2, norm="ortho")
        self._seed = itertools.count(0xC0FFEE + 1)    # (next() of a count is atomic: concurrent callers never draw the same dither seed)
        self._tables = {}

    def build(self, input_shape):
        M = input_shape[self.sampleAxis]
        self._prepare(M)
        self.built = True

    def _prepare(self, M):
        """(frame width) -> (FrontendCfg, host tables); shared by the stand-alone call and the fused extractor."""
        self.windowing.build((None, None, M))
        self.filterbank.build((None, None, M))
        self.dct.build((None, None, self.melBins))
        w, f = self.windowing, self.filterbank
        cfg = L.FrontendCfg(frame_size=M, frame_shift=M, nfft=max(64, f.fftLength), num_mels=self.melBins,
                            num_ceps=self.numMfccs, remove_dc=int(w.removeDCOffset), raw_energy=int(w.rawEnergy),
                            use_energy=int(self.useEnergy), use_power=int(f.usePower), use_log=int(f.useLogFBank),
                            use_lifter=int(self.cepstralLifter > 1 and self.lifters is not None), preemph=w.preemphasisCoeff,
                            dither=w.dither, energy_floor=w.energyFloor, eps=self.eps)
        self._M = M
        self._cfg = cfg
        return cfg

    def tables(self, device):
        key = (str(device), self._M)
        if key not in self._tables:
            self._tables[key] = ops.FrontendTables(self._M, window=self.windowing.windowFunc, mel_bank=self.filterbank.melBank,
                                                   dct=self.dct.dct, lifter=self.lifters, device=device)
        return self._tables[key]

    def compute_output_shape(self, input_shape):
        out = list(input_shape)
        out[self.sampleAxis] = self.numMfccs
        return out

    def get_config(self):
        c = super().get_config()
        c.update(self.windowing.get_config())
        c.update(self.filterbank.get_config())
        c.update({"num_mfccs": self.numMfccs, "num_mels": self.melBins, "cepstral_lifter": self.cepstralLifter,
                  "use_energy": self.useEnergy, "epsilon": self.eps})
        return c

    def next_seed(self):
        return next(self._seed)

    def call(self, inputs):
        x = inputs.to(torch.float32).contiguous()
        M = x.shape[-1]
        if M != self._M:
            raise ValueError(f"layer was built for frames of {self._M} samples, got {M}")
        B, T = x.shape[0], x.shape[-2]
        rows = x.numel() // M
        out = ops.frontend(x.reshape(1, rows, M), L.IN_FRAMES, self._cfg, self.tables(x.device), L.OUT_MFCC, rows, 1, rows,
                           seed=self.next_seed())
        return out.reshape(B, T, self.numMfccs)


class VAD(Layer):
    """layers/dsp/vad.py:23 — Kaldi energy VAD."""

    def __init__(self, energy_mean_scale=0.5, energy_threshold=5, frames_context=0, proportion_threshold=0.6,
                 return_indexes=True, energy_coeff=0, name=None, **kwargs):
        super().__init__(trainable=False, name=name, **kwargs)
        if energy_mean_scale < 0:
            raise ValueError("`energy_mean_scale` must be >= 0")
        if frames_context < 0:
            raise ValueError("`frames_context` must be >= 0")
        if proportion_threshold <= 0 or proportion_threshold >= 1:
            raise ValueError("`proportion_threshold` must be between 0 and 1 (exlcusive)")
        self.energyThreshold = float(energy_threshold)
        self.energyMeanScale = float(energy_mean_scale)
        self.propThreshold = float(proportion_threshold)
        self.returnIndexes = return_indexes
        self.useEnergyMean = energy_mean_scale > 0
        self.framesContext = frames_context
        self.windowSize = self.framesContext * 2 + 1
        self.energyCoef = energy_coeff
        self.frameAxis = -2

    def cfg(self):
        return L.VadCfg(energy_threshold=self.energyThreshold, energy_mean_scale=self.energyMeanScale,
                        proportion_threshold=self.propThreshold, frames_context=self.framesContext,
                        energy_coeff=self.energyCoef)

    def get_config(self):
        c = super().get_config()
        c.update({"energy_mean_scale": self.energyMeanScale, "energy_threshold": self.energyThreshold,
                  "frames_context": self.framesContext, "proportion_threshold": self.propThreshold,
                  "return_indexes": self.returnIndexes, "energy_coeff": self.energyCoef})
        return c

    def call(self, inputs):
        x = inputs.to(torch.float32).contiguous()
        if x.dim() == 2:
            x = x.unsqueeze(0)
        B, T, D = x.shape
        if self.returnIndexes:
            idx, lens = ops.vad_index(x, self.cfg())
            # tf.where row order: [batch, frame] ascending. The (n,2) shape is data dependent -> one host sync.
            lens_h = lens.cpu().tolist()
            rows = [torch.stack([torch.full((n,), b, dtype=torch.int64, device=x.device), idx[b, :n].to(torch.int64)], 1)
                    for b, n in enumerate(lens_h)]
            return torch.cat(rows, 0) if rows else torch.zeros((0, 2), dtype=torch.int64, device=x.device)
        return ops.vad_mask(x, self.cfg()).reshape(B, T, 1)


# =============================================================================== normalization
class CMVN(Layer):
    """layers/normalization/cmvn.py:25 — sliding-window cepstral mean (and variance) normalisation."""

    def __init__(self, center=True, norm_vars=False, window=600, min_window=100, padding="SAME", name=None, **kwargs):
        super().__init__(trainable=False, name=name)
        self.center = center
        self.normVar = norm_vars
        self.N = window
        self.minN = min_window
        if not self.center:
            raise NotImplementedError("CMVN with center=False not supported yet")
        if self.N <= 0 or self.minN <= 0:
            raise ValueError("`window` and `min_window` must be > 0")
        self.padding = padding.upper()
        if self.padding not in ["SAME", "VALID"]:
            raise ValueError(f"`padding` should be either 'SAME' or 'VALID', got '{padding}'")
        self.batchAxis, self.frameAxis, self.featAxis = 0, -2, -1

    def cfg(self):
        return L.CmvnCfg(window=self.N, norm_vars=int(self.normVar), valid=int(self.padding == "VALID"), reserved=0)

    def compute_output_shape(self, input_shape):
        if self.padding == "SAME":
            return input_shape
        out = list(input_shape)
        T = input_shape[self.frameAxis]
        if T is None:
            out[self.frameAxis] = None
        elif T <= self.N:
            out[self.frameAxis] = T
        else:
            out[self.frameAxis] = T - (2 * self.N - 1) // 2
        return out

    def get_config(self):
        c = super().get_config()
        c.update({"center": self.center, "norm_vars": self.normVar, "window": self.N, "min_window": self.minN,
                  "padding": self.padding})
        return c

    def call(self, inputs):
        x = inputs.to(torch.float32).contiguous()
        if x.dim() == 2:
            x = x.unsqueeze(0)
        B, T, D = x.shape
        out = ops.cmvn(x, self.cfg())
        if self.padding == "VALID":         # (cmvn.py:238-243: an input no longer than the window leaves no valid frame -- one when exactly as long)
            a, b = self.N // 2, T - (self.N - 1) // 2
            return out[:, : max(b - a, 0), :].contiguous()
        return out


class ReLU(Layer):
    """keras ReLU as used by models/kaldi/sequential.py:72 (stand-alone elementwise kernel)."""

    def __init__(self, name=None, **kwargs):
        super().__init__(trainable=False, name=name, **kwargs)

    def call(self, inputs):
        return ops.affine_act(inputs.to(torch.float32).contiguous(), L.ACT_RELU)


class BatchNorm(Layer):
    """layers/normalization/batchnorm.py:27 — keras BatchNormalization(center=False, scale=True) at inference with
    Kaldi weight import: y = gamma * (x - moving_mean) / sqrt(moving_var + eps)."""

    def __init__(self, axis=-1, momentum=0.99, target_rms=1.0, epsilon=0.001, mean_initializer=None,
                 variance_initializer=None, name=None, **kwargs):
        super().__init__(trainable=True, name=name, **kwargs)
        self.axis = axis
        self.momentum = momentum
        self.targetRMS = target_rms
        self.epsilon = epsilon
        self.gamma = self.moving_mean = self.moving_variance = None
        self._dev = None
        self._version = 0          # bumped by set_weights: consumers that folded this layer's affine into their weights re-fold

    def build(self, input_shape):
        D = input_shape[-1]
        if self.gamma is None:
            self.gamma = np.full((D,), self.targetRMS, np.float32)
            self.moving_mean = np.zeros((D,), np.float32)
            self.moving_variance = np.ones((D,), np.float32)
        self.built = True

    def get_config(self):
        c = super().get_config()
        c.update({"axis": self.axis, "momentum": self.momentum, "epsilon": self.epsilon, "target_rms": self.targetRMS})
        return c

    def get_weights(self):
        return [self.gamma, self.moving_mean, self.moving_variance]

    def set_weights(self, weights, fmt="kaldi"):
        if fmt not in ["kaldi", "tensorflow"]:
            raise ValueError(f"expected 'fmt' to be either 'kaldi' or 'tensorflow', got {fmt}")
        if len(weights) != 3:
            raise ValueError(f"expected a weight list of length 3, got {len(weights)}")
        if fmt == "tensorflow":
            gamma, mean, var = [np.asarray(w, np.float32) for w in weights]
        else:
            targetRMS, mean, var = weights
            mean = np.asarray(mean, np.float32)
            var = np.asarray(var, np.float32)
            gamma = (np.float32(targetRMS) * np.ones_like(mean)).astype(np.float32)
        if not (gamma.shape == mean.shape == var.shape):
            raise ValueError("gamma / mean / variance shapes differ")
        self.gamma, self.moving_mean, self.moving_variance = gamma, mean, var
        self._dev = None
        self._version += 1
        WEIGHTS_EPOCH[0] += 1

    def affine64(self):
        """(scale, shift) of the inference transform y = scale * x + shift, float64."""
        g = self.gamma.astype(np.float64)
        scale = g / np.sqrt(self.moving_variance.astype(np.float64) + self.epsilon)
        return scale, -self.moving_mean.astype(np.float64) * scale

    def affine(self):
        """(scale, shift) of the inference transform, computed in float64 on the host."""
        scale, shift = self.affine64()
        return scale.astype(np.float32), shift.astype(np.float32)

    def affine_device(self, device):
        if self._dev is None or self._dev[0].device != torch.device(device):
            s, h = self.affine()
            self._dev = (ops.to_device_f32(s, device), ops.to_device_f32(h, device))
        return self._dev

    def call(self, inputs, training=False):
        if training:
            raise NotImplementedError("training-mode batch normalisation is not part of the inference path")
        x = inputs.to(torch.float32).contiguous()
        s, h = self.affine_device(x.device)
        return ops.affine_act(x, L.ACT_NONE, s, h)


# =============================================================================== tdnn
def reshapeKaldiTdnnWeights(weights, units, kernel_width):
    """layers/tdnn/utils.py:22-28: Kaldi (units, K*D) -> keras conv2d kernel (1, K, D, units)."""
    return weights.flatten().reshape((1, -1, kernel_width, units), order="F").transpose([0, 2, 1, 3])


WEIGHTS_EPOCH = [0]      # bumped by every set_weights / re-build in the process: a cheap "anything changed?" for captured graphs

# every name tf.keras.activations.get resolves in the reference's TensorFlow (2.8; layers/tdnn/tdnn.py:117-118). The GEMM epilogues
# fuse the first four; the others run as a second launch over the layer's output (ktf_tdnn does that itself, fp32 kernels only)
_ACTS = {None: L.ACT_NONE, "linear": L.ACT_NONE, "relu": L.ACT_RELU, "sigmoid": L.ACT_SIGMOID, "tanh": L.ACT_TANH,
         "elu": L.ACT_ELU, "selu": L.ACT_SELU, "softplus": L.ACT_SOFTPLUS, "softsign": L.ACT_SOFTSIGN, "swish": L.ACT_SWISH,
         "gelu": L.ACT_GELU, "exponential": L.ACT_EXPONENTIAL, "hard_sigmoid": L.ACT_HARD_SIGMOID, "softmax": L.ACT_SOFTMAX}
_GEMM = {"f32": L.GEMM_F32, "float32": L.GEMM_F32, "bf16": L.GEMM_BF16, "bfloat16": L.GEMM_BF16, "bf16x3": L.GEMM_BF16X3,
         "f16mx": L.GEMM_F16MX}


class TDNN(Layer):
    """layers/tdnn/tdnn.py:29 — irregular-context 1-D convolution as an implicit-im2col MFMA GEMM."""

    MAX_DEVICE_SETS = 4        # device operand sets kept per layer AND DEVICE (keyed by mode / layout): the oldest set of
                               # the same device is evicted; sets on other devices (other ranks' streams may be reading them) stay

    def __init__(self, units, context=[0], subsampling_factor=1, padding="SAME", use_bias=True, kernel_initializer=None,
                 bias_initializer=None, activation=None, name=None, gemm="f32", **kwargs):
        super().__init__(trainable=True, name=name, **kwargs)
        self.units = units
        self.useBias = use_bias
        self.subsamplingFactor = subsampling_factor
        if self.subsamplingFactor <= 0:
            raise ValueError("subsampling_factor should be > 0")
        self.padding = padding.upper()
        if self.padding not in ["VALID", "SAME"]:
            raise ValueError("padding should be either 'VALID' or 'SAME'")
        if context is None:
            self.context = [0]
        elif isinstance(context, int):
            self.context = [context]
        elif isinstance(context, list):
            self.context = context if len(context) > 0 else [0]
        else:
            raise ValueError("context should be None, a list or an integer")
        self.context.sort()
        self.kernelWidth = len(self.context)
        self.kernelInitializer = kernel_initializer
        self.biasInitializer = bias_initializer
        self.activation = activation
        if isinstance(activation, str) and activation.lower() not in _ACTS:
            raise ValueError(f"Unknown activation function: {activation}")
        if gemm not in _GEMM:
            raise ValueError(f"gemm must be one of {sorted(_GEMM)}")
        self.gemm = gemm
        self.batchAxis, self.timeAxis, self.featAxis = 0, 1, -1
        self.kernel = None     # keras layout (1, K, D, units), numpy fp32
        self.bias = None
        self.kernelFlags = 0   # KtfTdnnDesc.flags of this layer's launches (L.TDNN_REF_TILES: bitwise-reference fp32 tiles)
        self._dev = {}
        self._version = 0      # bumped whenever the weights change: models refuse stale captured graphs

    # ---- weights
    def build(self, input_shape):
        D = input_shape[self.featAxis]
        self.inputDim = D
        if self.kernel is None or self.kernel.shape != (1, self.kernelWidth, D, self.units):
            # Glorot-uniform like the reference's default initialisers (tdnn.py:40-41)
            # seeded from the layer NAME with a process-independent hash: every rank of a multi-GPU job draws the same
            # initial weights for a layer that is not found in the nnet3 file
            rng = np.random.default_rng(zlib.crc32(self.name.encode()))
            fan_in, fan_out = self.kernelWidth * D, self.kernelWidth * self.units
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            self.kernel = rng.uniform(-lim, lim, (1, self.kernelWidth, D, self.units)).astype(np.float32)
            if self.useBias:
                lb = np.sqrt(6.0 / (self.units + 1))
                self.bias = rng.uniform(-lb, lb, (self.units,)).astype(np.float32)
            self._version += 1
            WEIGHTS_EPOCH[0] += 1
        self._dev = {}
        self.built = True

    def get_weights(self):
        return [self.kernel, self.bias] if self.useBias else [self.kernel]

    def set_weights(self, weights, fmt="kaldi"):
        fmt = fmt.lower()
        if fmt not in ["kaldi", "tensorflow"]:
            raise ValueError(f"expected 'fmt' to be either 'kaldi' or 'tensorflow', got {fmt}")
        if len(weights) == 0:
            raise ValueError("expected a weight list of at least length 2, got 0")
        if self.useBias and len(weights) != 2:
            raise ValueError(f"expected a weight list of length 2, got {len(weights)}")
        kernel = np.asarray(weights[0], np.float32)
        if fmt == "kaldi":
            kernel = reshapeKaldiTdnnWeights(kernel, self.units, self.kernelWidth)
        if self.built and kernel.shape != (1, self.kernelWidth, self.inputDim, self.units):
            raise ValueError(f"Layer weight shape {(1, self.kernelWidth, self.inputDim, self.units)} not compatible with "
                             f"provided weight shape {kernel.shape}")
        if kernel.ndim != 4 or kernel.shape[0] != 1 or kernel.shape[1] != self.kernelWidth or kernel.shape[3] != self.units:
            raise ValueError(f"weight shape {kernel.shape} does not match (1, {self.kernelWidth}, D, {self.units})")
        self.kernel = np.ascontiguousarray(kernel)
        self.inputDim = kernel.shape[2]
        if self.useBias:
            bias = np.asarray(weights[1], np.float32).reshape(-1)
            if bias.shape != (self.units,):
                raise ValueError(f"bias shape {bias.shape} != ({self.units},)")
            self.bias = bias
        self._dev = {}
        self._version += 1
        WEIGHTS_EPOCH[0] += 1
        self._warned_fp32_fallback = set()       # (warn_fallback speaks again for the new weights)
        self.built = True

    def kaldi_matrix(self):
        """(units, K*D) Kaldi layout of the kernel."""
        K, D = self.kernelWidth, self.inputDim
        return np.ascontiguousarray(self.kernel[0].reshape(K * D, self.units).T)

    def device_weights(self, device, gemm, k_interleaved=False, w_tiled=False):
        """Padded GEMM operands on the device: W (units_pad, K*Dpad) in the GEMM's dtype (+ lo part for the split-bf16 mode),
        bias. `k_interleaved`: K axis ordered (32-feature chunk, context, feature) — KTF_TDNN_K_INTERLEAVED, `w_tiled`: the
        kernel's LDS stage images (KTF_TDNN_W_TILED); both for the split-plane kernel."""
        key = (str(device), gemm, bool(k_interleaved), bool(w_tiled))
        if key in self._dev:
            return self._dev[key]
        K, D = self.kernelWidth, self.inputDim
        Dp, Up = ops.round_up(D, 32), ops.round_up(self.units, 256)
        Wk = np.transpose(self.kernel[0], (2, 0, 1)).astype(np.float64)       # [u, k, d]
        bias64 = self.bias.astype(np.float64) if self.useBias else None
        W = np.zeros((Up, K, Dp), np.float64)
        W[: self.units, :, :D] = Wk
        if k_interleaved:
            W = np.ascontiguousarray(W.reshape(Up, K, Dp // 32, 32).transpose(0, 2, 1, 3))
        W = W.reshape(Up, K * Dp)
        if w_tiled:
            # KTF_TDNN_W_TILED: (N-tile, K-step) blocks in the kernel's LDS image order: row r keeps its four 16-byte chunks
            # at positions chunk ^ ((4 - (r >> 2)) & 3)
            nt, nks = Up // 256, K * Dp // 32
            r = np.arange(256)
            src = np.arange(4)[None, :] ^ ((4 - ((r >> 2) & 3)) & 3)[:, None]              # [row, position] -> chunk
            W5 = W.reshape(nt, 256, nks, 4, 8)
            W = np.ascontiguousarray(W5[:, r[:, None], :, src, :].transpose(2, 3, 0, 1, 4)).reshape(Up, K * Dp)
        W = torch.as_tensor(W, device=device)
        w_lo = None
        if gemm == L.GEMM_F32:
            w = W.to(torch.float32)
        elif gemm == L.GEMM_BF16X4:
            w = ops.pair_encode(W.to(torch.float32))
        else:
            W = W.to(torch.float32)
            w = W.to(torch.bfloat16)
            if gemm == L.GEMM_BF16X3:
                w_lo = (W - w.to(torch.float32)).to(torch.bfloat16)
        bias = ops.to_device_f32(bias64, device) if bias64 is not None else None
        mine = [k for k in self._dev if k[0] == str(device)]
        while len(mine) >= self.MAX_DEVICE_SETS:             # mode changes replace operand sets: evict this device's oldest
            self._dev.pop(mine.pop(0))
        self._dev[key] = (w, w_lo, bias)
        return self._dev[key]

    def device_weights_mx(self, device, fold=None, loader=True, kernel=None):
        """KTF_GEMM_F16MX operands on the device: (wh, wq, bias) -- the half plane and the block-scaled e2m1 / e2m3 planes of
        the weights as the LDS images of the kernel that will read them (include/ktf_hip.h, ktf_tdnn_mx). `kernel`: "tile" = the
        256 x 256 kernel (mx.weight_images), "loader" = the loader-wave kernel (KTF_TDNN_MX_LOADER: mx.weight_images_loader);
        None: "loader" / "tile" by `loader`. K ordered (32-feature chunk, context, feature) and zero-padded to
        whole super-steps. `fold`: the BatchNorm whose affine y = s*x + h precedes this layer, folded INTO it so that the stored
        activations are the ReLU outputs themselves: W'[u,k,d] = W[u,k,d] * s[d], b'[u] = b[u] + sum_kd W[u,k,d] * h[d] (float64 on
        the host; exact for replicate padding, every context row carries the same per-feature affine)."""
        from . import mx
        kernel = kernel or ("loader" if loader else "tile")
        if kernel not in ("loader", "tile"):
            raise ValueError(f"unknown f16mx kernel {kernel!r}")
        key = ("mx", str(device), None if fold is None else (id(fold), fold._version), kernel)
        if key in self._dev:
            return self._dev[key]
        K, D = self.kernelWidth, self.inputDim
        Dp, Up = ops.round_up(D, 32), ops.round_up(self.units, 256)
        Wk = np.transpose(self.kernel[0], (2, 0, 1)).astype(np.float64)       # [u, k, d]
        bias64 = self.bias.astype(np.float64) if self.useBias else None
        if fold is not None:
            s64, h64 = fold.affine64()
            if s64.shape != (D,):
                raise ValueError(f"cannot fold a {s64.shape[0]}-wide BatchNorm into a layer with input dim {D}")
            extra = np.einsum("ukd,d->u", Wk, h64)
            bias64 = extra if bias64 is None else bias64 + extra
            Wk = Wk * s64[None, None, :]
        W = np.zeros((Up, K, Dp), np.float64)
        W[: self.units, :, :D] = Wk
        W = np.ascontiguousarray(W.reshape(Up, K, Dp // 32, 32).transpose(0, 2, 1, 3)).reshape(Up, (Dp // 32) * K, 32)
        wh, wq, _ = mx.weight_images_loader(W) if kernel == "loader" else mx.weight_images(W)
        out = (torch.as_tensor(wh, device=device), torch.as_tensor(wq, device=device),
               ops.to_device_f32(bias64, device) if bias64 is not None else None)
        # one MX image set per (layer, device, kernel): a re-fold replaces the set of its own device and kernel only
        self._dev = {k: v for k, v in self._dev.items() if not (k[0] == "mx" and k[1] == key[1] and k[3] == key[3])}
        self._dev[key] = out
        return out

    def desc(self, gemm, x_dtype, y_dtype, act=None, flags=0):
        d = L.TdnnDesc()
        d.flags = flags
        d.units, d.din, d.din_pad, d.nctx = self.units, self.inputDim, ops.round_up(self.inputDim, 32), self.kernelWidth
        for i, c in enumerate(self.context):
            d.ctx[i] = int(c)
        d.subsampling, d.valid = self.subsamplingFactor, int(self.padding == "VALID")
        a = self.activation if act is None else act
        d.act = _ACTS[a.lower() if isinstance(a, str) else a]
        d.gemm = gemm
        d.x_dtype = L.ktf_dtype(x_dtype)
        d.w_dtype = {L.GEMM_F32: L.KTF_F32, L.GEMM_BF16X4: L.KTF_BF16P}.get(gemm, L.KTF_BF16)
        d.y_dtype = L.ktf_dtype(y_dtype)
        return d

    def getStartEndSteps(self, T):
        start, end = 0, T
        if self.padding == "VALID":
            if self.context[0] < 0:
                start = -1 * self.context[0]
            if self.context[-1] > 0:
                end = T - self.context[-1]
        return start, end

    def outputTimesteps(self, T):
        start, end = self.getStartEndSteps(T)
        n = end - start
        return 0 if n <= 0 else (n + self.subsamplingFactor - 1) // self.subsamplingFactor

    def compute_output_shape(self, input_shape):
        B, T = input_shape[self.batchAxis], input_shape[self.timeAxis]
        return (B, None if T is None else self.outputTimesteps(T), self.units)

    def get_config(self):
        c = super().get_config()
        c.update({"units": self.units, "context": self.context, "subsampling_factor": self.subsamplingFactor,
                  "padding": self.padding, "use_bias": self.useBias, "activation": self.activation})
        return c

    def forward(self, x, lens=None, relu=False, bn=None, gemm=None, out_dtype=torch.float32, ldy=None, out=None,
                out_lens=None, flags=0, pair_in=False, pair_out=False):
        """Low-level launch used by call() and by the fused Sequential runner.
        x: (B, T, ldx) fp32/bf16 with ldx >= round_up(D,32) (pad columns finite). Returns y (B, Tout, ldy).
        `pair_in` (gemm GEMM_BF16X4) / `pair_out` (GEMM_F32 or GEMM_BF16X4): the float32 tensor x / y holds KTF_BF16P pairs."""
        gemm = _GEMM[self.gemm] if gemm is None else gemm
        if pair_in != (gemm == L.GEMM_BF16X4):
            raise ValueError("KTF_GEMM_BF16X4 reads pairs, and only it does")
        w, w_lo, bias = self.device_weights(x.device, gemm)
        act = "relu" if relu else None
        if relu and self.activation not in (None, "linear"):
            raise ValueError("cannot fuse a ReLU after a TDNN that already has an activation")
        d = self.desc(gemm, L.PAIR if pair_in else x.dtype, L.PAIR if pair_out else out_dtype, act=act if relu else None,
                      flags=0 if pair_in else flags | self.kernelFlags)
        B, T = x.shape[0], x.shape[1]
        Tout = self.outputTimesteps(T)
        ldy = self.units if ldy is None else ldy
        if out is None:
            alloc = torch.zeros if ldy != self.units else torch.empty
            out = alloc((B, Tout, ldy), dtype=out_dtype, device=x.device)
        scale, shift = bn if bn is not None else (None, None)
        if Tout > 0 and B > 0:
            ops.tdnn(x, lens, d, w, w_lo, bias, scale, shift, out, out_lens)
        elif out_lens is not None:                  # no output row (VALID padding of inputs shorter than the context): every utterance is empty
            out_lens.zero_()
        return out

    def effective_gemm(self, gemm, relu=False):
        """The f16mx mode runs on the MX kernels only (units > 128, ReLU or no activation); any other layer of
        such a model is evaluated by the exact fp32 kernel instead. A pure query (the fused runner's planner asks it for every
        layer, and for the layer behind it): `warn_fallback` is what tells the user."""
        return L.GEMM_F32 if self.fallback_reason(gemm, relu) else gemm

    def fallback_reason(self, gemm, relu=False):
        """Why this layer runs on the exact fp32 kernels in a model of mode `gemm` (None: it does not)."""
        a = self.activation.lower() if isinstance(self.activation, str) else self.activation
        if gemm == L.GEMM_F32:
            return None
        if _ACTS[a] > L.ACT_TANH:            # activations no epilogue fuses: the fp32 kernels + an activation pass, in every mode
            return f"activation {self.activation!r} is not fused by any reduced-precision epilogue"
        if gemm != L.GEMM_F16MX:
            return None
        if self.units <= 128:
            return f"units = {self.units} (these kernels tile 256 units: layers of up to 128 stay on fp32)"
        if not (a in (None, "linear") or (a == "relu" and not relu)):
            return (f"activation {self.activation!r}" + (" followed by a ReLU" if relu else "") + " (these kernels fuse ReLU or no activation)")
        return None

    def warn_fallback(self, gemm, relu=False):
        """One RuntimeWarning per (layer, mode) naming the condition that keeps the layer off the mode's kernels; called where the
        layer is about to run (TDNN.call, the fused runner), re-armed by set_weights."""
        why = self.fallback_reason(gemm, relu)
        if why is None or gemm in self.__dict__.setdefault("_warned_fp32_fallback", set()):
            return
        import warnings
        self._warned_fp32_fallback.add(gemm)
        mode = {v: k for k, v in _GEMM.items()}.get(gemm, str(gemm))
        warnings.warn(f"TDNN layer '{self.name}' runs on the exact fp32 kernels in this '{mode}' model: {why} "
                      f"(several times slower per flop; its results are the more accurate ones)", RuntimeWarning, stacklevel=3)

    def prepare_input(self, x, gemm):
        """Dense user tensor (B,T,D) -> operand the GEMM can read (padded to a multiple of 32 columns, right dtype)."""
        D = x.shape[-1]
        Dp = ops.round_up(D, 32)
        want = L.act_torch_dtype(gemm)
        if D == Dp and x.dtype == want and x.is_contiguous():
            return x
        if x.dtype not in (torch.float32, torch.bfloat16) or (x.dtype != torch.float32 and x.dtype != want):
            x = x.to(torch.float32)
        dst = torch.empty((*x.shape[:-1], Dp), dtype=want, device=x.device)
        return ops.convert_pad(x.contiguous(), D, dst)

    def call(self, inputs):
        x = inputs
        if x.dim() != 3:
            raise ValueError(f"expected a (batch, time, feat) input, got shape {tuple(x.shape)}")
        if x.shape[-1] != self.inputDim:
            raise ValueError(f"expected input feature dim {self.inputDim}, got {x.shape[-1]}")
        self.warn_fallback(_GEMM[self.gemm])
        gemm = self.effective_gemm(_GEMM[self.gemm])
        B, T, D = x.shape
        if gemm == L.GEMM_F16MX:
            return self._call_mx(x)
        # the 16-bit ring kernels want an output row stride that is a multiple of 8 (16-byte stores); the pad columns
        # are sliced off again
        ldy = ops.round_up(self.units, 8) if gemm == L.GEMM_BF16 else self.units
        if T == 1 and list(self.context) == [0] and self.subsamplingFactor == 1 and B > 1:
            # one row per utterance (e.g. the affine after stats pooling): run as ONE B-row GEMM (context [0] only: another offset
            # would clamp to the neighbouring UTTERANCES' rows there, not to the utterance's single frame)
            y = self.forward(self.prepare_input(x.reshape(1, B, D), gemm), gemm=gemm, ldy=ldy)
            return y[:, :, : self.units].reshape(B, 1, self.units)
        return self.forward(self.prepare_input(x, gemm), gemm=gemm, ldy=ldy)[:, :, : self.units]


    def _call_mx(self, x):
        """A stand-alone call of an "f16mx" layer (inside a Sequential the planes travel from layer to layer instead): fp32 rows ->
        the four MX planes -> csrc/tdnn_mx.hip -> fp32 rows."""
        from . import mx
        B, T, D = x.shape
        src = x if (x.dtype == torch.float32 and x.stride(2) == 1 and x.stride(0) == T * x.stride(1)) else x.to(torch.float32).contiguous()
        planes = mx.Planes.empty(B, T, D, x.device)
        ops.mx_planes(src, D, None, planes)
        wh, wq, bias = self.device_weights_mx(x.device, loader=False)
        d = self.desc(L.GEMM_F16MX, torch.float32, torch.float32)
        if self.outputTimesteps(T) <= 0:               # VALID padding of an input shorter than the context: no output row
            return torch.empty((B, 0, self.units), dtype=torch.float32, device=x.device)
        y = torch.empty((B, self.outputTimesteps(T), ops.round_up(self.units, 4)), dtype=torch.float32, device=x.device)
        ops.tdnn_mx(planes, None, d, wh, wq, bias, None, None, y)
        return y[:, :, : self.units]


# =============================================================================== stats
class StatsPooling(Layer):
    """layers/stats/stats_pooling.py:26 — mean (+ std) pooling over the whole utterance or sliding windows."""

    def __init__(self, left_context, right_context, input_period=1, output_period=1, include_std=True, padding="SAME",
                 epsilon=1e-10, reduce_time_axis=False, name=None, **kwargs):
        super().__init__(trainable=False, name=name)
        self.leftContext = left_context
        self.rightContext = right_context
        self.inputPeriod = input_period
        self.outputPeriod = output_period
        self.includeStd = include_std
        self.reduce = reduce_time_axis
        if self.leftContext > 0 or self.rightContext < 0:
            raise ValueError("'left_context' must be <= 0 and 'right_context' must be >= 0")
        if self.inputPeriod <= 0 or self.outputPeriod <= 0:
            raise ValueError("'input_period' and 'output_period' must be > 0")
        if self.outputPeriod % self.inputPeriod != 0 and not self.reduce:
            raise ValueError("'output_period' must be a multiple of 'input_period'")
        self.padding = padding.upper()
        if self.padding not in ["VALID", "SAME"]:
            raise ValueError("padding should be either 'VALID' or 'SAME'")
        self.epsilon = float(epsilon)
        self.maxWindowWidth = right_context - left_context + 1
        self.batchAxis, self.timeAxis, self.featAxis = 0, 1, -1

    def getStartEndSteps(self, T):
        start, end = 0, T
        if self.padding == "SAME":
            return start, end
        if self.leftContext < 0:
            start = -1 * self.leftContext
        if self.rightContext > 0 and self.maxWindowWidth < T:
            end = T - self.rightContext
        return start, end + 1

    def numOutputSteps(self, T):
        start, end = self.getStartEndSteps(T)
        n = end - start
        return 0 if n <= 0 else (n + self.outputPeriod - 1) // self.outputPeriod

    def compute_output_shape(self, input_shape):
        B, T, D = input_shape[self.batchAxis], input_shape[self.timeAxis], input_shape[self.featAxis]
        if self.includeStd:
            D = D * 2
        if self.reduce:
            return (B, 1, D)
        if self.padding == "SAME" or T is None:          # (a model built for any number of frames: the time axis stays unknown)
            return (B, T, D)
        return (B, self.numOutputSteps(T), D)

    def get_config(self):
        c = super().get_config()
        c.update({"left_context": self.leftContext, "right_context": self.rightContext, "input_period": self.inputPeriod,
                  "output_period": self.outputPeriod, "include_std": self.includeStd, "padding": self.padding,
                  "epsilon": self.epsilon, "reduce_time_axis": self.reduce})
        return c

    def reduce_all(self, x, D, lens=None, out=None):
        """(B,T,ld) -> (B, 1, D or 2D) over the valid rows of each utterance."""
        od = 2 * D if self.includeStd else D
        if out is None:
            out = torch.empty((x.shape[0], od), dtype=torch.float32, device=x.device)
        ops.stats_pool(x, D, lens, self.inputPeriod, self.includeStd, self.epsilon, out)
        return out

    def call(self, inputs):
        x = inputs if inputs.dtype in (torch.float32, torch.bfloat16) else inputs.to(torch.float32)
        x = x.contiguous()
        B, T, D = x.shape
        od = 2 * D if self.includeStd else D
        if self.reduce:
            return self.reduce_all(x, D).reshape(B, 1, od)
        if T == 0 and self.padding == "SAME":        # no frame, no window (a VALID-padded layer in front left nothing); with VALID
            return torch.empty((B, 0, od), dtype=torch.float32, device=x.device)      # padding the reference pools "all" frames: one NaN row, below
        x = x.to(torch.float32)
        if self.padding == "SAME":
            n = self.numOutputSteps(T)
            s = ops.stats_pool_windowed(x, self.leftContext, self.rightContext, self.inputPeriod, self.outputPeriod, 0, n,
                                        self.includeStd, self.epsilon)
            if self.outputPeriod > 1:
                # tf.repeat(stats, output_period, axis=time): row j -> rows j*p .. j*p+p-1 (pure data movement)
                s = s.repeat_interleave(self.outputPeriod, dim=1)
            return s
        if T > self.maxWindowWidth:
            start, _ = self.getStartEndSteps(T)
            return ops.stats_pool_windowed(x, self.leftContext, self.rightContext, self.inputPeriod, self.outputPeriod,
                                           start, self.numOutputSteps(T), self.includeStd, self.epsilon)
        return self.reduce_all(x, D).reshape(B, 1, od)


# =============================================================================== plda
class PLDA(Layer):
    """layers/plda/plda.py:24 — PLDA transform + pairwise log-likelihood-ratio scoring."""

    def __init__(self, dim, plda_mean, plda_transform, plda_psi, normalize_length=True, simple_length_norm=False,
                 dtype=torch.float64, return_transformed=True, name=None):
        super().__init__(trainable=False, name=name)
        self.dim = dim
        self.normalizeLength = normalize_length
        self.simpleLengthNorm = simple_length_norm
        self.paramDtype = self._torch_dtype(dtype)
        self.returnTransformed = return_transformed
        self.inputRank = 3
        npdt = np.float64 if self.paramDtype == torch.float64 else np.float32
        self.mean = np.asarray(plda_mean, dtype=npdt)
        self.transformMat = np.asarray(plda_transform, dtype=npdt)
        self.psi = np.asarray(plda_psi, dtype=npdt)
        self.assertParamShapes()
        self.offset = (-1.0 * (self.transformMat @ self.mean.reshape(self.dim, 1))).reshape(-1).astype(npdt)
        self._dev = None

    @staticmethod
    def _torch_dtype(dt):
        if dt in (torch.float64, np.float64, "float64", float):
            return torch.float64
        if dt in (torch.float32, np.float32, "float32"):
            return torch.float32
        name = getattr(dt, "name", None)
        if name in ("float64", "float32"):
            return torch.float64 if name == "float64" else torch.float32
        raise ValueError(f"unsupported PLDA dtype {dt}")

    def build(self, input_shape):
        vecDim = input_shape[-1]
        if vecDim != self.dim:
            raise ValueError(f"expected input vector dimension to be {self.dim}, got {vecDim}")
        self.inputRank = len(input_shape)
        if self.inputRank not in [2, 3]:
            raise ValueError(f"expected input tensor rank to be 2 or 3, got {len(input_shape)}")
        self.built = True

    def assertParamShapes(self):
        assert self.mean.ndim == 1, f"plda_mean must be a vector, got dimension={self.mean.ndim}"
        assert self.psi.ndim == 1, f"plda_psi must be a vector, got dimension={self.psi.ndim}"
        assert self.transformMat.ndim == 2, f"plda_transform_mat must be a matrix, got dimension={self.transformMat.ndim}"
        assert self.mean.shape[0] == self.dim, f"plda_mean dimension size ({self.mean.shape[0]}) != input dim ({self.dim})"
        assert self.psi.shape[0] == self.dim, f"plda_psi dimension size ({self.psi.shape[0]}) != input dim ({self.dim})"
        r, c = self.transformMat.shape
        assert r == self.dim, f"plda_transform_mat dimension size ({r}) != input dim ({self.dim})"
        assert r == c, f"plda_transform_mat ({r} x {c}) is not a square matrix"

    def _prepare(self, inputs):
        x = inputs
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.ascontiguousarray(x), device=ops.default_device())
        if x.dim() == 3:
            if x.shape[1] != 1:
                raise ValueError(f"expected (batch, 1, dim) input, got {tuple(x.shape)}")
            x = x.reshape(x.shape[0], x.shape[2])
        elif x.dim() != 2:
            raise ValueError(f"expected input tensor rank to be 2 or 3, got {x.dim()}")
        x = x.to(self.paramDtype).contiguous()
        if self._dev is None or self._dev[0].device != x.device:
            f = lambda a: torch.as_tensor(np.ascontiguousarray(a).copy(), device=x.device)  # noqa: E731
            self._dev = (f(self.transformMat), f(self.offset), f(self.psi))
        return x

    def transform(self, inputs):
        """Extension: transformVector alone (plda.py:163-196) -> (B, dim) transformed vectors."""
        x = self._prepare(inputs)
        A, off, psi = self._dev
        return ops.plda(x, A, off, psi, self.normalizeLength, self.simpleLengthNorm, want_scores=False)[1]

    def score(self, test_transformed, enroll_transformed):
        """Extension: rectangular trial block, scores[i, j] = LLR(test_i | class of enroll_j) on TRANSFORMED vectors
        (the reference scores a batch against itself only); row blocks of a large trial matrix shard across GPUs
        (parallel.plda_trials)."""
        t = test_transformed.reshape(test_transformed.shape[0], -1).to(self.paramDtype).contiguous()
        e = enroll_transformed.reshape(enroll_transformed.shape[0], -1).to(self.paramDtype).contiguous()
        if self._dev is None or self._dev[0].device != t.device:
            self._prepare(t)
        return ops.plda_score(t, e, self._dev[2])

    def call(self, inputs):
        x = self._prepare(inputs)
        A, off, psi = self._dev
        scores, tr = ops.plda(x, A, off, psi, self.normalizeLength, self.simpleLengthNorm)
        if self.returnTransformed:
            return scores, tr.reshape(tr.shape[0], tr.shape[1], 1)
        return scores
