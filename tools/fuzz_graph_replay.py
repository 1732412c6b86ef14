#!/usr/bin/env python3
"""A captured extractor graph replayed on inputs it was not captured on (GPU box): compile once per (mode, batch shape) on stationary noise,
then replay on batches mixing speech, bursts followed by silence, quiet noise and digital silence -- the voiced lengths, and with them the
per-utterance routing of the f16mx model, change from replay to replay while the launches do not. Every replay must equal the eager call
on the same input bit for bit (NaN where the eager call gives NaN).   python tools/fuzz_graph_replay.py [rounds] [seed]"""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
import kaldi_tflite_amd as ktf

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=77)
speech = synth.speech_wavs()[0][0]
bad = 0
for mode in ("f16mx", "bf16x3", "f32"):
    for (B, N) in ((1, 160000), (3, 80000), (8, 48000 + 13), (40, 160000)):
        mdl = synth.build_extractor(ktf, cfg, w, gemm=mode)
        first = torch.as_tensor(synth.make_wav(B, N, seed=5), device="cuda")
        run = mdl.compile(first)
        for r in range(rounds):
            wav = np.zeros((B, N), np.float32)
            for b in range(B):
                k = int(rng.integers(0, 5))
                if k == 0:
                    wav[b] = synth.make_wav(1, N, seed=int(rng.integers(1 << 30)), ragged=True)[0]
                elif k == 1:
                    o = int(rng.integers(0, len(speech) - N))
                    wav[b] = speech[o:o + N]
                elif k == 2:
                    n = int(rng.integers(800, min(N, 40000)))
                    wav[b, :n] = synth.make_wav(1, n, seed=int(rng.integers(1 << 30)))[0]
                elif k == 3:
                    wav[b] = synth.make_wav(1, N, seed=int(rng.integers(1 << 30)), sigma=3.0)[0]
            x = torch.as_tensor(wav, device="cuda")
            a = run(x).float().cpu().numpy().reshape(B, -1)
            e = mdl(x).float().cpu().numpy().reshape(B, -1)
            same = np.array_equal(np.isnan(a), np.isnan(e)) and np.array_equal(a[~np.isnan(a)], e[~np.isnan(e)])
            if not same:
                bad += 1
                d = np.nanmax(np.abs(a - e)) if np.isfinite(a).any() and np.isfinite(e).any() else float("nan")
                print(f"MISMATCH {mode} B {B} N {N} replay {r}: graph != eager (max difference {d:.3e}, NaN rows {np.isnan(a).any(1).sum()} / {np.isnan(e).any(1).sum()})", flush=True)
print(f"{rounds} replays per graph, {bad} mismatches")
sys.exit(min(bad, 255))
