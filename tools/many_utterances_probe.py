#!/usr/bin/env python3
"""Deviation of the f16x2 x-vectors from the fp64 oracle over 32 ten-second utterances (16 all voiced, 16 with quiet blocks):
per-utterance max-abs and the rms over all components, calibrated default and two passes everywhere (checker; GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O
ktf.models.Sequential.min_tiles = {}
cfg = synth.extractor_cfg()
w = synth.make_weights(seed=4321)
N = 160000
wav = np.concatenate([synth.make_wav(16, N, seed=9001), synth.make_wav(16, N, seed=9002, ragged=True)], 0)
want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
for cal in (True, False):
    m = synth.build_extractor(ktf, cfg, w, gemm="f16x2", calibrate=cal)
    got = m(torch.as_tensor(wav, device="cuda")).cpu().numpy()
    e = np.abs(got - want).max(1)
    print(f"calibrated {cal}: per-utterance max-abs x1e-5: min {e.min()*1e5:.2f} median {np.median(e)*1e5:.2f} max {e.max()*1e5:.2f}; rms over all components {np.sqrt(((got-want)**2).mean())*1e5:.2f}")
