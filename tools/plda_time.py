#!/usr/bin/env python3
"""BASELINE config 5: the 1024 x 1024 PLDA trial matrix (dim 128) on the x-vector side of the path, ms per call (transform + scores),
fp64 and fp32: python tools/plda_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
import kaldi_tflite_amd as ktf
rng = np.random.default_rng(31)
dim, nb = 128, 1024
A = rng.standard_normal((dim, dim)) / np.sqrt(dim) + np.eye(dim)
mean, psi = rng.standard_normal(dim) * 0.1, np.sort(rng.uniform(0.05, 30.0, dim))[::-1].copy()
for dt in (torch.float64, torch.float32):
    plda = ktf.layers.PLDA(dim, mean, A, psi, dtype=dt)
    xv = torch.as_tensor(rng.standard_normal((nb, dim)), device="cuda").to(dt)
    t = plda.transform(xv)
    print(str(dt), "transform + scores", round(bench._time_ms(torch, lambda: plda(xv), 20), 4), "ms; scores alone",
          round(bench._time_ms(torch, lambda: plda.score(t, t), 20), 4), "ms")
