"""
TEST INFRASTRUCTURE / CPU BASELINE ONLY — never imported by the product (kaldi_tflite_amd); only tests/ and
bench.py's `cpu_baseline` leg may use it.

`ktf_ref`: a torch-CPU fp32 restatement of the reference's wav -> x-vector op graph, written to mirror the COST
STRUCTURE of the reference's TensorFlow-CPU path (SURVEY.md §8d "CPU baseline") so that it can be timed
next to the GPU path on the GPU box's host cores:

  Framing      materialised (B,T,400) frames           (tf.gather of frame indexes, layers/dsp/framing.py:243-265)
  Windowing    elementwise passes over the frames      (layers/dsp/windowing.py:180-209)
  FilterBank   zero-pad, torch.fft.rfft, |.|^2, DENSE (257,30) mel matmul, log      (layers/dsp/filterbank.py:225-242)
  DCT/MFCC     (30,30) matmul, lifter, C0 <- log-energy                             (layers/dsp/mfcc.py:197-244)
  VAD          mean threshold, 5-tap count, edge denominators, compaction           (layers/dsp/vad.py:156-203)
  CMVN         difference of fp32 cumulative sums                                   (layers/normalization/cmvn.py:146-250)
  TDNN         MATERIALISED im2col (index_select of the context rows) + matmul      (layers/tdnn/tdnn.py:251-280)
  ReLU, BatchNorm (eps 1e-3), StatsPooling (mean | std), tdnn6, mean/LDA/length-norm
                                                                                    (xvector_extractor.py:174-184)

All constants (window, mel bank, DCT, lifter, BatchNorm affine, transposed weights) are computed ONCE in __init__ from the
NumPy oracle's table functions; `tests/test_oracle_golden.py` checks this module against `ktf_oracle.xvector_forward`.
The reference is defined for batch 1 only (it concatenates the voiced frames of a batch); a batch here is B independent
utterances evaluated with batched tensors when their voiced-frame counts agree (the stationary synthetic workload) and
utterance by utterance otherwise.
"""

import numpy as np
import torch

from . import ktf_oracle as O


class KtfRef:
    def __init__(self, cfg, layers, global_mean, lda_mat):
        f = cfg["framing"]
        self.size, self.shift = O.frame_params(f["frame_length_ms"], f["frame_shift_ms"], f["sample_frequency"])[:2]
        self.size = 2 * (self.size // 2)
        m = dict(cfg["mfcc"])
        self.num_ceps, self.num_mels = m.get("num_mfccs", 23), m.get("num_mels", 23)
        self.preemph = float(m.get("preemphasis_coefficient", 0.97))
        self.remove_dc = bool(m.get("remove_dc_offset", True))
        self.raw_energy = bool(m.get("raw_energy", True))
        self.energy_floor = float(m.get("energy_floor", 0.0))
        self.eps = float(m.get("epsilon", 1e-7))
        self.use_energy = bool(m.get("use_energy", True))
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
        self.window = t(O.window_function(m.get("window_type", "povey"), self.size))
        self.nfft, bank = O.mel_bank(self.size, self.num_mels, m.get("sample_frequency", 16000.0),
                                     m.get("high_freq_cutoff", 0.0), m.get("low_freq_cutoff", 20.0))
        self.mel = t(bank)                                                 # (nfft/2+1, mels), dense like the reference
        self.dct = t(O.dct_matrix(self.num_mels, self.num_ceps))
        q = m.get("cepstral_lifter", 22)
        self.lifter = t(O.lifter_coeffs(self.num_ceps, q)) if q > 1 else None
        self.vad = dict(cfg["vad"])
        self.cmvn_window = int(cfg["cmvn"].get("window", 600))
        self.cmvn_norm_vars = bool(cfg["cmvn"].get("norm_vars", False))
        self.steps = []
        for l in layers:
            if l["kind"] == "tdnn":
                W = np.asarray(l["W"], np.float32)
                self.steps.append(("tdnn", t(W.T), t(l["b"]) if l.get("b") is not None else None, list(l["context"])))
            elif l["kind"] == "relu":
                self.steps.append(("relu",))
            elif l["kind"] == "bn":
                scale = np.float64(l["rms"]) / np.sqrt(np.asarray(l["var"], np.float64) + 1e-3)
                self.steps.append(("bn", t(scale), t(-np.asarray(l["mean"], np.float64) * scale)))
            elif l["kind"] == "stats":
                self.steps.append(("stats",))
            else:
                raise ValueError(l["kind"])
        lda = np.asarray(lda_mat, np.float32)
        self.mean = t(global_mean)
        self.lda = t(lda[:, :-1].T)
        self.lda_off = t(lda[:, -1])

    # ------------------------------------------------------------------ front-end
    def mfcc(self, wav):
        """(B,N) -> (B,T,ceps): materialised frames, rfft, dense mel matmul."""
        frames = wav.unfold(-1, self.size, self.shift).contiguous()                    # (B,T,size) copy, as tf.gather
        if self.remove_dc:
            frames = frames - frames.mean(-1, keepdim=True)
        log_e = None
        if self.use_energy and self.raw_energy:
            log_e = torch.log(frames.pow(2).sum(-1).clamp_min(0) + self.eps).clamp_min(self.energy_floor)
        if self.preemph > 0:
            prev = torch.cat([frames[..., :1], frames[..., :-1]], -1)                  # split/concat copy of the reference
            frames = frames - self.preemph * prev
        frames = frames * self.window
        if self.use_energy and not self.raw_energy:
            log_e = torch.log(frames.pow(2).sum(-1).clamp_min(0) + self.eps).clamp_min(self.energy_floor)
        spec = torch.fft.rfft(frames, n=self.nfft)                                     # zero-padded to nfft
        power = spec.abs().pow(2)
        fb = torch.log((power @ self.mel).clamp_min(0) + self.eps)
        c = fb @ self.dct
        if self.lifter is not None:
            c = c * self.lifter
        if self.use_energy:
            c = torch.cat([log_e.unsqueeze(-1), c[..., 1:]], -1)
        return c

    def vad_keep(self, m):
        """(B,T,C) -> bool (B,T)."""
        v = self.vad
        e = m[..., int(v.get("energy_coeff", 0))]
        thr = float(v.get("energy_threshold", 5.0))
        if v.get("energy_mean_scale", 0.5) > 0:
            thr = thr + float(v["energy_mean_scale"]) * e.mean(-1, keepdim=True)
        d = (e > thr).to(torch.float32)
        ctx = int(v.get("frames_context", 0))
        if ctx == 0:
            return d > 0
        T = d.shape[-1]
        cnt = torch.nn.functional.conv1d(d.unsqueeze(1), torch.ones(1, 1, 2 * ctx + 1), padding=ctx).squeeze(1)
        den = torch.full((T,), float(2 * ctx + 1))
        for j in range(ctx):                                                            # vad.py:124-135 edge sizes
            den[j % T] = ctx + 1 + j
        for j in range(ctx):
            den[(T - ctx + j) % T] = 2 * ctx - j
        return (cnt / den) >= float(v.get("proportion_threshold", 0.6))

    def cmvn(self, x):
        """(B,T,C) sliding-window mean (variance) normalisation from cumulative sums."""
        N, T = self.cmvn_window, x.shape[-2]
        if T <= N:
            y = x - x.mean(-2, keepdim=True)
            return y / x.std(-2, unbiased=False, keepdim=True) if self.cmvn_norm_vars else y
        start = (torch.arange(T) - N // 2).clamp(0, T - N)
        cs = torch.nn.functional.pad(x, (0, 0, 1, 0)).cumsum(-2)
        mean = (cs.index_select(-2, start + N) - cs.index_select(-2, start)) / N
        y = x - mean
        if self.cmvn_norm_vars:
            cs2 = torch.nn.functional.pad(x * x, (0, 0, 1, 0)).cumsum(-2)
            var = (cs2.index_select(-2, start + N) - cs2.index_select(-2, start)) / N - mean * mean
            y = y / var.sqrt()
        return y

    # ------------------------------------------------------------------ network
    def network(self, x):
        """(B,T,C) -> (B,512): materialised im2col + matmul per TDNN layer."""
        for st in self.steps:
            if st[0] == "tdnn":
                _, Wt, b, ctx = st
                T = x.shape[-2]
                if T > 1 or len(ctx) > 1:
                    t = torch.arange(T)
                    cols = [x.index_select(-2, (t + c).clamp(0, T - 1)) for c in ctx]   # tf.gather: (B,T,K,D) materialised
                    x = torch.cat(cols, -1)
                x = x @ Wt
                if b is not None:
                    x = x + b
            elif st[0] == "relu":
                x = torch.relu(x)
            elif st[0] == "bn":
                x = x * st[1] + st[2]
            else:
                mean = x.mean(-2, keepdim=True)
                var = ((x * x).mean(-2, keepdim=True) - mean * mean).clamp_min(0)
                x = torch.cat([mean, (var + 1e-10).sqrt()], -1)
        return x.squeeze(-2)

    def post(self, h):
        y = (h - self.mean) @ self.lda + self.lda_off
        return y * (float(y.shape[-1]) ** 0.5) / y.norm(dim=-1, keepdim=True)

    @torch.no_grad()
    def features(self, wav):
        """wav (B,N) -> CMVN'd MFCCs of all frames (no VAD): the 'Framing + MFCC + CMVN' configuration of BASELINE.json."""
        return self.cmvn(self.mfcc(torch.as_tensor(wav, dtype=torch.float32)))

    @torch.no_grad()
    def __call__(self, wav):
        wav = torch.as_tensor(wav, dtype=torch.float32)
        m = self.mfcc(wav)
        keep = self.vad_keep(m)
        counts = keep.sum(-1)
        if int(counts.min()) == int(counts.max()):
            n = int(counts[0])
            voiced = m[keep].reshape(m.shape[0], n, m.shape[-1])                        # gather_nd
            return self.post(self.network(self.cmvn(voiced)))
        return torch.cat([self.post(self.network(self.cmvn(m[b:b + 1][keep[b:b + 1]].unsqueeze(0)))) for b in range(m.shape[0])], 0)
