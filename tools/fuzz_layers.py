#!/usr/bin/env python3
"""Differential fuzzing of the layer API against the oracle (GPU box): random configurations of StatsPooling, CMVN, TDNN, VAD, Framing
and MFCC including degenerate lengths (inputs shorter than windows / contexts). Prints every mismatch; exit code = their number.
Test infrastructure: the oracle is the checker, nothing here is on the product path.   python tools/fuzz_layers.py [n_cases] [seed]"""
import os, sys, warnings
warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import layers as Ls
from oracle import ktf_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def report(kind, cfg, msg):
    global bad
    bad += 1
    print(f"MISMATCH {kind} {cfg}: {msg}", flush=True)


def compare(kind, cfg, got, want, tol):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    if got.shape != want.shape:
        return report(kind, cfg, f"shape {got.shape} != {want.shape}")
    if want.size:
        nan_w, nan_g = ~np.isfinite(want), ~np.isfinite(got)      # 0 / 0 of the reference (one-frame variances, empty windows): the same places
        if not np.array_equal(nan_w, nan_g):
            return report(kind, cfg, f"non-finite values in {nan_g.sum()} places, the oracle in {nan_w.sum()}")
        ok = ~nan_w
        if ok.any():
            err = np.abs(got[ok] - want[ok]).max() / max(1.0, np.abs(want[ok]).max())
            if not err <= tol:
                report(kind, cfg, f"max deviation {err:.3e} > {tol}")


def guarded(kind, cfg, fn_got, fn_want, tol):
    try:
        want = fn_want()
    except Exception as e:                      # the reference raises: so must we
        try:
            fn_got()
            report(kind, cfg, f"oracle raises {type(e).__name__}: {e}; the layer does not")
        except Exception:
            pass
        return
    try:
        got = fn_got()
    except Exception as e:
        return report(kind, cfg, f"layer raises {type(e).__name__}: {e}")
    compare(kind, cfg, got, want, tol)


for case in range(n_cases):
    # ---- StatsPooling
    T, D, B = int(rng.choice([1, 2, 3, 7, 20, 64, 150])), int(rng.choice([5, 32, 40])), int(rng.integers(1, 4))
    lc, rc = -int(rng.integers(0, 12)), int(rng.integers(0, 12))
    ip = int(rng.choice([1, 1, 2, 3]))
    op = ip * int(rng.choice([1, 1, 2, 3]))
    cfg = dict(left_context=lc, right_context=rc, input_period=ip, output_period=op, include_std=bool(rng.integers(0, 2)),
               padding=str(rng.choice(["SAME", "VALID"])), reduce_time_axis=bool(rng.random() < 0.25))
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    guarded("StatsPooling", dict(cfg, T=T, D=D), lambda: Ls.StatsPooling(**cfg)(dev(x)), lambda: O.stats_pooling(x, **cfg, dtype=np.float64), 2e-5)
    # ---- CMVN
    T = int(rng.choice([1, 5, 50, 99, 100, 101, 150, 333]))
    N = int(rng.choice([10, 100, 101, 300]))
    cfg = dict(window=N, norm_vars=bool(rng.integers(0, 2)), padding=str(rng.choice(["SAME", "VALID"])))
    x = (rng.standard_normal((B, T, D)) * 2 + 1).astype(np.float32)
    guarded("CMVN", dict(cfg, T=T, D=D), lambda: Ls.CMVN(**cfg)(dev(x)), lambda: O.cmvn(x, **cfg, dtype=np.float64), 2e-4)
    # ---- TDNN (fp32 and the reduced modes at layer level)
    T = int(rng.choice([1, 2, 5, 9, 40, 257, 300]))
    K = int(rng.integers(1, 5))
    ctx = sorted(set(int(v) for v in rng.integers(-6, 7, K)))
    U = int(rng.choice([16, 130, 256, 300]))
    Din = int(rng.choice([24, 64, 100]))
    sub = int(rng.choice([1, 1, 2, 3]))
    pad = str(rng.choice(["SAME", "VALID"]))
    act = rng.choice([None, "relu", "tanh"])
    gemm, tol = [("f32", 3e-5), ("bf16x3", 3e-4), ("f16mx", 3e-3)][int(rng.integers(0, 3))]
    x = rng.standard_normal((B, T, Din)).astype(np.float32)
    W = (rng.standard_normal((U, len(ctx) * Din)) / np.sqrt(len(ctx) * Din)).astype(np.float32)
    b = rng.standard_normal(U).astype(np.float32)
    cfg = dict(T=T, D=Din, U=U, ctx=ctx, sub=sub, pad=pad, act=act, gemm=gemm)

    def run_tdnn():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t = Ls.TDNN(U, context=list(ctx), subsampling_factor=sub, padding=pad, activation=act, gemm=gemm)
            t.build(x.shape)
            t.set_weights([W, b])
            return t(dev(x))
    guarded("TDNN", cfg, run_tdnn, lambda: O.tdnn(x, W, b, ctx, sub, pad, act, dtype=np.float64), tol)
    # ---- VAD (mask and index forms)
    T = int(rng.choice([1, 2, 3, 4, 5, 9, 100]))
    vc = dict(energy_mean_scale=float(rng.choice([0.0, 0.5])), energy_threshold=float(rng.choice([5.0, 5.5])),
              frames_context=int(rng.integers(0, 4)), proportion_threshold=float(rng.choice([0.12, 0.6])))
    f = (rng.standard_normal((1, T, 30)) * 4 + 6).astype(np.float32)
    for ri in (False, True):
        guarded("VAD", dict(vc, T=T, return_indexes=ri), lambda: Ls.VAD(**vc, return_indexes=ri)(dev(f)).to(torch.float64),
                lambda: np.asarray(O.vad(f, **vc, return_indexes=ri), np.float64), 0.0)
    # ---- Framing + MFCC
    sf = float(rng.choice([8000.0, 16000.0]))
    fl, fs = float(rng.choice([20.0, 25.0, 32.0])), float(rng.choice([10.0, 12.5]))
    n = int(rng.choice([0, 1, 100, int(sf * fl / 1000) - 1, int(sf * fl / 1000), int(sf * fl / 1000) + 1, 3000, 8000]))
    wav = (rng.standard_normal((B, n)) * 1000).astype(np.float32)
    fc = dict(frame_length_ms=fl, frame_shift_ms=fs, sample_frequency=sf)
    guarded("Framing", dict(fc, n=n), lambda: Ls.Framing(**fc)(dev(wav)), lambda: O.framing(wav, **fc).astype(np.float64), 0.0)
    if n >= int(sf * fl / 1000):
        nm = int(rng.choice([23, 30, 40]))
        mc = dict(num_mfccs=int(rng.integers(5, nm + 1)), num_mels=nm, cepstral_lifter=float(rng.choice([0.0, 22.0])),
                  use_energy=bool(rng.integers(0, 2)), sample_frequency=sf, low_freq_cutoff=float(rng.choice([20.0, 100.0])),
                  high_freq_cutoff=float(rng.choice([0.0, -200.0, 3700.0])), window_type=str(rng.choice(["povey", "hamming", "hanning", "rectangular", "blackman"])),
                  remove_dc_offset=bool(rng.integers(0, 2)), preemphasis_coefficient=float(rng.choice([0.0, 0.97])), raw_energy=bool(rng.integers(0, 2)))
        frames = O.framing(wav, **fc)
        guarded("MFCC", dict(mc, **fc, n=n), lambda: Ls.MFCC(**mc)(dev(frames)), lambda: O.mfcc(frames, **mc, dtype=np.float64), 2e-3)
    # ---- Windowing / FilterBank / DCT on random frames
    M = int(rng.choice([200, 256, 400, 512]))
    fr = (rng.standard_normal((B, int(rng.integers(1, 6)), M)) * 300).astype(np.float32)
    wc = dict(window_type=str(rng.choice(["povey", "hamming", "hanning", "rectangular", "blackman"])), blackman_coeff=float(rng.choice([0.42, 0.3])),
              remove_dc_offset=bool(rng.integers(0, 2)), preemphasis_coefficient=float(rng.choice([0.0, 0.5, 0.97, 1.0])),
              raw_energy=bool(rng.integers(0, 2)), energy_floor=float(rng.choice([0.0, 1.0])), return_energy=True)
    try:
        gw, ge = Ls.Windowing(**wc)(dev(fr))
        ww, we = O.windowing(fr, **wc, dtype=np.float64)
        compare("Windowing", dict(wc, M=M), gw, ww, 1e-5)
        compare("Windowing.energy", dict(wc, M=M), ge, we, 1e-5)
    except Exception as e:
        report("Windowing", dict(wc, M=M), f"raises {type(e).__name__}: {e}")
    sfb = float(rng.choice([8000.0, 16000.0]))
    bc = dict(num_bins=int(rng.choice([10, 23, 40, 64])), sample_frequency=sfb, low_freq_cutoff=float(rng.choice([0.0, 20.0, 300.0])),
              high_freq_cutoff=float(rng.choice([0.0, -100.0, sfb / 2 - 500])), use_log_fbank=bool(rng.integers(0, 2)), use_power=bool(rng.integers(0, 2)))
    guarded("FilterBank", dict(bc, M=M), lambda: Ls.FilterBank(**bc)(dev(fr)), lambda: O.filterbank(fr, **bc, dtype=np.float64), 3e-4)
    nin = int(rng.choice([10, 23, 40]))
    nout = int(rng.integers(1, nin + 1))
    z = rng.standard_normal((B, 7, nin)).astype(np.float32)
    guarded("DCT", dict(nin=nin, nout=nout), lambda: Ls.DCT(nout)(dev(z)), lambda: O.dct(z, nout, dtype=np.float64), 1e-5)
    # ---- PLDA
    dim = int(rng.choice([5, 29, 64, 128, 200]))
    nb = int(rng.choice([1, 2, 9, 70]))
    mean = rng.standard_normal(dim)
    A = rng.standard_normal((dim, dim)) / np.sqrt(dim)
    psi = rng.uniform(0.1, 5.0, dim)
    xv = rng.standard_normal((nb, dim))
    pc = dict(normalize_length=bool(rng.integers(0, 2)), simple_length_norm=bool(rng.integers(0, 2)))
    for dt, tolp in ((np.float64, 1e-9), (np.float32, 2e-3)):
        def run_plda():
            sc, tr = Ls.PLDA(dim, mean, A, psi, dtype=dt, **pc)(dev(xv.astype(dt)))
            return sc
        guarded("PLDA", dict(pc, dim=dim, B=nb, dtype=dt.__name__), run_plda, lambda: O.plda(xv, mean, A, psi, **pc, dtype=np.float64)[0], tolp)
    # ---- Framing with snip_edges=False == framing of the mirror-padded waveform (kaldi_numpy.PadWaveform)
    if n >= 2:
        fsz, fsh, _ = O.frame_params(**fc)
        guarded("Framing(snip_edges=False)", dict(fc, n=n), lambda: Ls.Framing(**fc, snip_edges=False)(dev(wav)),
                lambda: O.framing(O.pad_waveform(wav, fsz, fsh), **fc).astype(np.float64), 0.0)
print(f"{n_cases} rounds, {bad} mismatches")
sys.exit(min(bad, 255))
