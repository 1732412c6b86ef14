#!/usr/bin/env python3
"""Latency of one 10 s utterance (BASELINE config 2) per GEMM mode: eager launches and the captured hipGraph, ms per call, and the
x-vector's deviation from the fp64 oracle. `python tools/batch1_latency.py [mode ...]` (default: f32 f16mx bf16x3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
import bench
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O

modes = sys.argv[1:] or ["f32", "f16mx", "bf16x3"]
dev = torch.device("cuda", 0)
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
wav_h = synth.make_wav(1, 160000, seed=3)
want = O.xvector_forward(wav_h, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
wav = torch.as_tensor(wav_h, device=dev)
for mode in modes:
    for pairs in ((True, False) if mode != "f32" else (True,)):
        m = synth.build_extractor(ktf, cfg, w, gemm=mode)
        m.xvec.small_tile_pairs = pairs
        err = float(np.abs(m(wav).cpu().numpy() - want).max())
        eager = bench._time_ms(torch, lambda: m(wav), 200)
        run = m.compile(wav)
        graph = bench._time_ms(torch, lambda: run(wav), 200)
        print(f"{mode:7s} pairs={pairs!s:5s} eager {eager:.4f} ms  hipgraph {graph:.4f} ms  max-abs dev vs fp64 oracle {err:.2e}")
