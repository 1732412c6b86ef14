#!/usr/bin/env python3
"""Differential fuzzing of XvectorExtractor (wav -> x-vector, full-size 0008 topology, synthetic weights) against the fp64 oracle on the
GPU box: batches that mix long, short (per-utterance routing), partly silent and completely silent utterances, fp32 and int16 input,
every compliant mode (f32 / bf16x3 / f16mx), default routing, eager and captured. The bar is north_star's: max-abs deviation <= 1e-4;
an utterance without a voiced frame must come out NaN as in the reference (0 / 0 in the pooling).
Test infrastructure: the oracle is the checker.   python tools/fuzz_extractor.py [rounds] [seed]"""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
KNOBS = "--knobs" in sys.argv        # ... with random settings of the runner's A/B knobs per round (kernel choices, fusion, routing): every
                                      # combination must stay inside the tolerance -- they change schedules and kernels, not the arithmetic contract
RANDOM_CFG = "--cfg" in sys.argv     # ... with a random front-end per round as well (sampling rate, frame length / shift, mel / cepstrum sizes and
                                      # options, VAD and CMVN settings: the generic front-end kernel beside the nfft-512 one), narrow network


def random_cfg():
    sf = float(rng.choice([8000.0, 16000.0]))
    nm = int(rng.choice([23, 30, 40]))
    nc = int(rng.choice([13, 23, nm])) if nm >= 23 else nm
    c = synth.extractor_cfg()
    c["framing"].update({"frame_length_ms": float(rng.choice([20.0, 25.0, 32.0])), "frame_shift_ms": float(rng.choice([10.0, 12.5])), "sample_frequency": sf})
    c["mfcc"].update({"num_mfccs": min(nc, nm), "num_mels": nm, "sample_frequency": sf, "high_freq_cutoff": float(rng.choice([0.0, -200.0, sf / 2 - 400.0])),
                      "low_freq_cutoff": float(rng.choice([20.0, 100.0])), "cepstral_lifter": float(rng.choice([0.0, 22.0])),
                      "use_energy": bool(rng.integers(0, 2)), "raw_energy": bool(rng.integers(0, 2)), "remove_dc_offset": bool(rng.integers(0, 2)),
                      "preemphasis_coefficient": float(rng.choice([0.0, 0.97])), "window_type": str(rng.choice(["povey", "hamming", "hanning", "blackman", "rectangular"]))})
    c["vad"].update({"frames_context": int(rng.integers(0, 4)), "proportion_threshold": float(rng.choice([0.12, 0.5])),
                     "energy_mean_scale": float(rng.choice([0.0, 0.5])), "energy_threshold": float(rng.choice([5.0, 5.5, 7.0]))})
    c["cmvn"].update({"window": int(rng.choice([100, 300, 301])), "norm_vars": bool(rng.integers(0, 2))})
    return c


cfg = synth.extractor_cfg()
w = synth.make_weights(seed=4321 + seed)
olayers = synth.oracle_layers(w)
models = {m: synth.build_extractor(ktf, cfg, w, gemm=m) for m in ("f32", "bf16x3", "f16mx")} if not RANDOM_CFG else {}
speech = synth.speech_wavs()[0][0]
bad = 0
seen = {"utterances": 0, "nan_expected": 0, "skipped_rounds": 0}
worst = {m: 0.0 for m in ("f32", "bf16x3", "f16mx")}
for r in range(rounds):
    if RANDOM_CFG:
        cfg = random_cfg()
        w = synth.make_weights(seed=int(rng.integers(1 << 30)), narrow=True, feat_dim=cfg["mfcc"]["num_mfccs"], out_dim=int(rng.choice([64, 128])))
        olayers = synth.oracle_layers(w)
        models = {m: synth.build_extractor(ktf, cfg, w, gemm=m) for m in ("f32", "bf16x3")}
        worst.update({m: worst.get(m, 0.0) for m in models})
    B = int(rng.choice([1, 2, 3, 5]))
    N = int(rng.choice([16000 * 2, 16000 * 3 + 77, 16000 * 5, 16000 * 7 + 5, 16000 * 10, 400, 1200, 16000]))
    wav = np.zeros((B, N), np.float32)
    kinds = []
    for b in range(B):
        k = str(rng.choice(["noise", "ragged", "speech", "short_then_silence", "silence", "quiet"]))
        kinds.append(k)
        if k == "noise":
            wav[b] = synth.make_wav(1, N, seed=int(rng.integers(1 << 30)))[0]
        elif k == "ragged":
            wav[b] = synth.make_wav(1, N, seed=int(rng.integers(1 << 30)), ragged=True)[0]
        elif k == "speech":
            o = int(rng.integers(0, len(speech) - N))
            wav[b] = speech[o:o + N]
        elif k == "short_then_silence":                  # a burst at the head, digital silence behind it: few voiced frames
            n = min(N, int(rng.integers(800, 8000)))
            wav[b, :n] = synth.make_wav(1, n, seed=int(rng.integers(1 << 30)))[0]
        elif k == "quiet":
            wav[b] = synth.make_wav(1, N, seed=int(rng.integers(1 << 30)), sigma=3.0)[0]
        # "silence": zeros
    try:
        want, inter = O.xvector_forward(wav, cfg, olayers, w["mean"], w["lda"], dtype=np.float64, return_intermediates=True)
        voiced = [len(i["voiced"]) for i in inter]
    except Exception as e:
        print(f"round {r}: oracle raises ({type(e).__name__}: {e}) for kinds {kinds}, N {N}; skipped")
        seen["skipped_rounds"] += 1
        continue
    # conditioning of the case: the oracle's own fp32 evaluation against its fp64 one (a handful of voiced frames under norm_vars, or a
    # pooled standard deviation over three frames, amplifies ANY fp32 rounding); a deviation counts when it exceeds both 1e-4 and 10 x that
    try:
        w32 = O.xvector_forward(wav, cfg, olayers, w["mean"], w["lda"], dtype=np.float32)
        cond = np.array([np.nanmax(np.abs(w32[b] - want[b])) if np.isfinite(want[b]).all() and np.isfinite(w32[b]).all() else np.inf for b in range(B)])
    except Exception:
        cond = np.zeros(B)
    seen["utterances"] += B
    seen["nan_expected"] += int(np.isnan(want).any(axis=1).sum())
    knobs = {}
    if KNOBS:
        knobs = {"mx_loader": [None, True, False][int(rng.integers(0, 3))], "flat_rows": bool(rng.integers(0, 2)), "mx_flat_rows": bool(rng.integers(0, 2)), "flat_pooling": bool(rng.integers(0, 2)),
                 "split_planes": bool(rng.integers(0, 2)), "fuse_stats": bool(rng.integers(0, 2)), "deterministic": bool(rng.integers(0, 2)),
                 "small_tile_pairs": bool(rng.integers(0, 2)), "min_tiles": [None, {}][int(rng.integers(0, 2))]}
        xk = {"fuse_tail": bool(rng.integers(0, 2)), "route_short_utterances": True}
        for mdl in models.values():
            for k, v in knobs.items():
                if k == "min_tiles":
                    mdl.xvec.min_tiles = dict(mdl.xvec.MIN_TILES) if v is None else {}
                elif hasattr(mdl.xvec, k):
                    setattr(mdl.xvec, k, v)
            for k, v in xk.items():
                setattr(mdl, k, v)
            mdl._graphs = {}
        knobs.update(xk)
    for mode, mdl in models.items():
        for form in ("fp32", "int16", "graph"):
            x = torch.as_tensor(wav if form != "int16" else wav.astype(np.int16), device="cuda")
            try:
                got = (mdl.compile(x)(x) if form == "graph" else mdl(x)).float().cpu().numpy().reshape(B, -1)
            except Exception as e:
                bad += 1
                print(f"MISMATCH round {r} {mode} {form} kinds {kinds} N {N} voiced {voiced}: raises {type(e).__name__}: {e}" + (f" knobs {knobs}" if KNOBS else ""), flush=True)
                continue
            for b in range(B):
                wn, gn = np.isnan(want[b]).any(), np.isnan(got[b]).any()
                if wn != gn:
                    bad += 1
                    print(f"MISMATCH round {r} {mode} {form} utterance {b} ({kinds[b]}, {voiced[b]} voiced frames, N {N}): NaN {gn}, oracle {wn}", flush=True)
                elif not wn:
                    err = np.abs(got[b] - want[b]).max()
                    worst[mode] = max(worst[mode], float(err))
                    if mode == "f32" and form == "fp32" and err > 1e-5:
                        print(f"note: round {r} f32 utterance {b} ({kinds[b]}, {voiced[b]} voiced frames, N {N}): {err:.2e} (oracle fp32 vs fp64: {cond[b]:.1e})", flush=True)
                    if not err <= max(1e-4, 10.0 * cond[b]):
                        bad += 1
                        print(f"MISMATCH round {r} {mode} {form} utterance {b} ({kinds[b]}, {voiced[b]} voiced frames, N {N}): max-abs deviation {err:.3e} (oracle fp32 vs fp64: {cond[b]:.1e})"
                              + (f" cfg {cfg}" if RANDOM_CFG else "") + (f" knobs {knobs}" if KNOBS else ""), flush=True)
print(f"{rounds} rounds, {bad} mismatches")
print(seen, {m: f"{v:.2e}" for m, v in worst.items()})
sys.exit(min(bad, 255))
