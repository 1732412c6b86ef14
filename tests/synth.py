"""Synthetic workloads shared by the tests, __graft_entry__.smoke() and bench.py:
0008_sitw_v2_1a topology with seeded random weights (the pretrained final.raw is not shipped with the
reference and there is no network), the real mean.vec / transform.mat fixtures for the LDA step, and
stationary-noise 16 kHz waveforms in int16 scale. Also a tiny Kaldi nnet3 *binary writer* (test tool) so that
SequentialFromConfig(cfg, nnet3Path) / XvectorExtractorFromConfig are exercised through the real file format."""

import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TOPOLOGY = [  # data/kaldi_models/configs/0008_sitw_v2_1a.yml:12-43 of the reference
    ("tdnn1", [-2, -1, 0, 1, 2], "h"), ("tdnn2", [-2, 0, 2], "h"), ("tdnn3", [-3, 0, 3], "h"),
    ("tdnn4", [0], "h"), ("tdnn5", [0], "p"),
]


def extractor_cfg(dither=0.0):
    """data/tflite_models/0008_sitw_v2_1a.yml:28-63 with dither 0 (the yml's 1.0 makes the output random)."""
    return {
        "framing": {"frame_length_ms": 25, "frame_shift_ms": 10, "sample_frequency": 16000, "dynamic_input_shape": True},
        "mfcc": {"num_mfccs": 30, "num_mels": 30, "sample_frequency": 16000.0, "high_freq_cutoff": 7600.0,
                 "low_freq_cutoff": 20.0, "dither": dither},
        "vad": {"energy_mean_scale": 0.5, "energy_threshold": 5.5, "frames_context": 2, "proportion_threshold": 0.12,
                "return_indexes": True, "energy_coeff": 0},
        "cmvn": {"center": True, "norm_vars": False, "window": 300},
    }


def extractor_cfg_8k():
    """Front-end of the reference's second shipped model, data/kaldi_models/configs/0006_callhome_diarization_v2_1a.yml
    (8 kHz, 23-dim MFCC: 200-sample frames -> nfft 256, i.e. the generic front-end kernel, not the nfft-512 fast path)."""
    c = extractor_cfg()
    c["framing"]["sample_frequency"] = 8000
    c["mfcc"].update({"num_mfccs": 23, "num_mels": 23, "sample_frequency": 8000.0, "high_freq_cutoff": 3700.0})
    return c


def model_config(narrow=False, feat_dim=30, out_dim=512):
    h, p = (64, 96) if narrow else (512, 1500)
    dims = {"h": h, "p": p}
    layers = [{"name": "input", "type": "input", "shape": [None, None, feat_dim]}]
    for name, ctx, d in TOPOLOGY:
        layers.append({"name": name, "type": ["affine", "relu", "batchnorm"], "cfg": {"units": dims[d], "context": list(ctx)}})
    layers.append({"name": "stats", "type": "stats_pooling",
                   "cfg": {"left_context": 0, "right_context": 10000, "include_std": True, "reduce_time_axis": True}})
    layers.append({"name": "tdnn6", "type": "affine", "cfg": {"units": out_dim, "context": [0]}})
    return {"type": "sequential", "layers": layers}


def make_weights(seed=4321, narrow=False, feat_dim=30, out_dim=512, bn="default", tails="normal"):
    """W ~ N(0, 1/(K*D)), b ~ N(0, 0.1), BN mean ~ U(0.2,1), var ~ U(0.5,2), target-rms 1; real LDA/mean
    (synthetic mean / LDA when the embedding is not 512-dimensional). Variants for the margin tests (the default draws are
    unchanged): bn="wide": BatchNorm var ~ U(0.05, 4), mean ~ U(-0.5, 1.5) (units far from unit scale, dead and hot ones);
    tails="t3": W from a Student t with 3 degrees of freedom scaled to the same variance (a few large weights per row, the rest
    smaller: what the block-scaled e2m1 / e2m3 images of a weight row like least)."""
    rng = np.random.default_rng(seed)
    h, p = (64, 96) if narrow else (512, 1500)
    dims = {"h": h, "p": p}
    w = {"feat_dim": feat_dim, "out_dim": out_dim}
    din = feat_dim
    for name, ctx, d in TOPOLOGY:
        u, K = dims[d], len(ctx)
        W = rng.standard_normal((u, K * din))
        if tails == "t3":
            W = rng.standard_t(3.0, (u, K * din)) / np.sqrt(3.0)          # (variance of t_3 is 3)
        w[f"{name}.affine"] = ((W / np.sqrt(K * din)).astype(np.float32), (rng.standard_normal(u) * 0.1).astype(np.float32))
        if bn == "wide":
            w[f"{name}.batchnorm"] = (np.float32(1.0), rng.uniform(-0.5, 1.5, u).astype(np.float32), rng.uniform(0.05, 4.0, u).astype(np.float32))
        else:
            w[f"{name}.batchnorm"] = (np.float32(1.0), rng.uniform(0.2, 1.0, u).astype(np.float32),
                                      rng.uniform(0.5, 2.0, u).astype(np.float32))
        din = u
    w["tdnn6.affine"] = ((rng.standard_normal((out_dim, 2 * din)) / np.sqrt(2 * din)).astype(np.float32),
                         (rng.standard_normal(out_dim) * 0.1).astype(np.float32))
    if out_dim == 512:
        w["mean"] = _read_text_vec(os.path.join(GOLDEN, "xvectors_train_combined_200k.mean.vec.txt"))
        w["lda"] = _read_bin_mat(os.path.join(GOLDEN, "xvectors_train_combined_200k.transform.mat"))
    else:
        w["mean"] = (rng.standard_normal(out_dim) * 0.1).astype(np.float32)
        w["lda"] = (rng.standard_normal((out_dim // 2, out_dim + 1)) / np.sqrt(out_dim)).astype(np.float32)
    w["narrow"] = narrow
    return w


def _read_text_vec(path):
    toks = open(path).read().replace("[", " ").replace("]", " ").split()
    return np.array([float(t) for t in toks], np.float32)


def _read_bin_mat(path):
    raw = open(path, "rb").read()
    assert raw[:2] == b"\0B" and raw[2:5] == b"FM "
    r = struct.unpack("<i", raw[6:10])[0]
    c = struct.unpack("<i", raw[11:15])[0]
    return np.frombuffer(raw[15:15 + 4 * r * c], np.float32).reshape(r, c).copy()


def oracle_layers(w):
    """Layer dicts for oracle.ktf_oracle.sequential_forward."""
    L = []
    for name, ctx, _ in TOPOLOGY:
        W, b = w[f"{name}.affine"]
        rms, mean, var = w[f"{name}.batchnorm"]
        L += [{"kind": "tdnn", "W": W, "b": b, "context": list(ctx)}, {"kind": "relu"},
              {"kind": "bn", "rms": rms, "mean": mean, "var": var}]
    L.append({"kind": "stats", "left_context": 0, "right_context": 10000, "include_std": True, "reduce_time_axis": True})
    W, b = w["tdnn6.affine"]
    L.append({"kind": "tdnn", "W": W, "b": b, "context": [0]})
    return L


def build_sequential(ktf, w, gemm="f32"):
    mdl = ktf.models.SequentialFromConfig(model_config(w["narrow"], w.get("feat_dim", 30), w.get("out_dim", 512)), None,
                                          "cmvn2xvec", gemm=gemm)
    for layer in mdl.layers:
        if layer.name in w:
            layer.set_weights(list(w[layer.name]))
    return mdl


def build_extractor(ktf, cfg, w, gemm="f32"):
    return ktf.models.XvectorExtractor.from_parts(cfg, build_sequential(ktf, w, gemm), w["mean"], w["lda"])


def make_wav(B, N, seed=1234, sigma=1000.0, ragged=False):
    """round(sigma * N(0,1)) clipped to int16, fp32 storage. `ragged`: 30 % of the 0.5 s blocks are scaled by 1e-3
    so the energy VAD drops frames (different counts per utterance)."""
    rng = np.random.default_rng(seed)
    x = np.clip(np.round(sigma * rng.standard_normal((B, N))), -32767, 32767).astype(np.float32)
    if ragged:
        blk = 8000
        nb = (N + blk - 1) // blk
        quiet = rng.random((B, nb)) < 0.3
        quiet[:, 0] = False
        g = np.repeat(np.where(quiet, 1e-3, 1.0), blk, axis=1)[:, :N].astype(np.float32)
        x = np.round(x * g).astype(np.float32)
    return x


def second_speech_wav():
    """The reference's OTHER recording: the 3 s clip its feature tests run on (testdata/feats/src/fbank_mfcc/16000_001/audio.wav,
    committed as tests/golden/feats_fbank_mfcc.npz:wav_int16), fp32 in int16 scale, (48000,)."""
    return np.load(os.path.join(GOLDEN, "feats_fbank_mfcc.npz"))["wav_int16"].astype(np.float32)


def speech_wavs(n=160000):
    """The reference's end-to-end test input (testdata/librispeech_2.wav, 22.5 s of read speech at 16 kHz; the input of
    models/kaldi/xvector_extractor_test.py:70-96, committed as tests/golden/e2e_0008.npz:wav_int16) as fp32 in int16 scale:
    (whole recording (1, 359665), its first two `n`-sample chunks (2, n)). Non-stationary audio: the VAD drops the pauses and
    the activation statistics move along the utterance -- what the stationary noise of make_wav cannot exercise."""
    z = np.load(os.path.join(GOLDEN, "e2e_0008.npz"))
    sp = z["wav_int16"].astype(np.float32)
    return sp[None, :], np.stack([sp[:n], sp[n:2 * n]], 0)


def coloured_am_noise(B, N, seed=99):
    """1/f-coloured noise with a 3 Hz / 0.31 Hz amplitude modulation (int16 scale): between stationary white noise and speech."""
    rng = np.random.default_rng(seed)
    f = np.fft.rfftfreq(N, 1 / 16000.0)
    t = np.arange(N) / 16000.0
    out = []
    for _ in range(B):
        pink = np.fft.irfft(np.fft.rfft(rng.standard_normal(N)) / np.sqrt(np.maximum(f, 20.0)), N)
        pink *= 1000.0 / pink.std()
        out.append(np.round(pink * (0.55 + 0.45 * np.sin(2 * np.pi * 3.0 * t)) * (1 + 0.5 * np.sin(2 * np.pi * 0.31 * t))))
    return np.clip(np.stack(out, 0), -32767, 32767).astype(np.float32)


# ----------------------------------------------------------------------------- nnet3 binary writer (test tool)
def _tok(s):
    return s.encode() + b" "


def _i32(v):
    return b"\x04" + struct.pack("<i", int(v))


def _f32(v):
    return b"\x04" + struct.pack("<f", float(v))


def _f64(v):
    return b"\x08" + struct.pack("<d", float(v))


def _fv(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    return b"FV " + _i32(a.size) + a.tobytes()


def _fm(a):
    a = np.ascontiguousarray(a, np.float32)
    return b"FM " + _i32(a.shape[0]) + _i32(a.shape[1]) + a.tobytes()


def write_nnet3(path, w):
    """Writes a binary <Nnet3> raw model with the layout KaldiNnet3Reader parses (observed in the reference's
    testdata/tdnn/src/tdnn_narrow/final.raw)."""
    names = [n for n, _, _ in TOPOLOGY]
    cfg = ["input-node name=input dim=30"]
    prev = "input"
    comps = []
    for name, ctx, _ in TOPOLOGY:
        app = ", ".join(prev if c == 0 else f"Offset({prev}, {c})" for c in ctx)
        inp = f"Append({app})" if len(ctx) > 1 else prev
        cfg.append(f"component-node name={name}.affine component={name}.affine input={inp}")
        cfg.append(f"component-node name={name}.relu component={name}.relu input={name}.affine")
        cfg.append(f"component-node name={name}.batchnorm component={name}.batchnorm input={name}.relu")
        prev = f"{name}.batchnorm"
    cfg.append("component-node name=tdnn6.affine component=tdnn6.affine input=stats-pooling-0-10000")
    cfg.append("output-node name=output input=tdnn6.affine")
    out = b"\0B" + b"<Nnet3> \n" + "\n".join(cfg).encode() + b"\n\n"
    for name in names:
        W, b = w[f"{name}.affine"]
        rms, mean, var = w[f"{name}.batchnorm"]
        dim = W.shape[0]
        comps.append(_tok("<ComponentName>") + _tok(f"{name}.affine") + _tok("<NaturalGradientAffineComponent>")
                     + _tok("<MaxChange>") + _f32(0.75) + _tok("<LinearParams>") + _fm(W) + _tok("<BiasParams>") + _fv(b)
                     + _tok("<RankIn>") + _i32(20) + _tok("</NaturalGradientAffineComponent>"))
        comps.append(_tok("<ComponentName>") + _tok(f"{name}.relu") + _tok("<RectifiedLinearComponent>")
                     + _tok("<Dim>") + _i32(dim) + _tok("<ValueAvg>") + _fv([]) + _tok("<DerivAvg>") + _fv([])
                     + _tok("<Count>") + _f64(0) + _tok("<OderivRms>") + _fv([]) + _tok("<OderivCount>") + _f64(0)
                     + _tok("</RectifiedLinearComponent>"))
        comps.append(_tok("<ComponentName>") + _tok(f"{name}.batchnorm") + _tok("<BatchNormComponent>")
                     + _tok("<Dim>") + _i32(dim) + _tok("<BlockDim>") + _i32(dim) + _tok("<Epsilon>") + _f32(1e-3)
                     + _tok("<TargetRms>") + _f32(rms) + _tok("<TestMode>") + b"F " + _tok("<Count>") + _f64(1e6)
                     + _tok("<StatsMean>") + _fv(mean) + _tok("<StatsVar>") + _fv(var) + _tok("</BatchNormComponent>"))
    W, b = w["tdnn6.affine"]
    comps.append(_tok("<ComponentName>") + _tok("stats-extraction-0-10000") + _tok("<StatisticsExtractionComponent>")
                 + _tok("<InputDim>") + _i32(W.shape[1] // 2) + _tok("</StatisticsExtractionComponent>"))
    comps.append(_tok("<ComponentName>") + _tok("stats-pooling-0-10000") + _tok("<StatisticsPoolingComponent>")
                 + _tok("<InputDim>") + _i32(W.shape[1] // 2 + 1) + _tok("</StatisticsPoolingComponent>"))
    comps.append(_tok("<ComponentName>") + _tok("tdnn6.affine") + _tok("<NaturalGradientAffineComponent>")
                 + _tok("<LinearParams>") + _fm(W) + _tok("<BiasParams>") + _fv(b) + _tok("</NaturalGradientAffineComponent>"))
    out += _tok("<NumComponents>") + _i32(len(comps)) + b"".join(comps) + _tok("</Nnet3>")
    with open(path, "wb") as f:
        f.write(out)


def write_text_vec(path, v):
    with open(path, "w") as f:
        f.write(" [ " + " ".join(repr(float(x)) for x in np.asarray(v).reshape(-1)) + " ]\n")


def write_bin_mat(path, m):
    with open(path, "wb") as f:
        f.write(b"\0B" + _fm(m))
