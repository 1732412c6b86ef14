#!/usr/bin/env python3
"""Does the half MFMA honour subnormal weight residuals? One F16X2 layer against fp64 with / without the lo part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L, ops
rng = np.random.default_rng(0)
for sigma in (1.0, 0.09, 0.0255, 0.005):
    B, T, D, U = 2, 300, 512, 512
    x = rng.standard_normal((B, T, D)).astype(np.float16)
    W = (rng.standard_normal((U, D)) * sigma).astype(np.float32)
    t = ktf.layers.TDNN(U, context=[0], gemm="f16x2", use_bias=False)
    t.build((B, T, D)); t.set_weights([W])
    w, w_lo, _ = t.device_weights("cuda:0", L.GEMM_F16X2)
    xd = torch.as_tensor(x, device="cuda")
    y = torch.zeros((B, T, 512), dtype=torch.float32, device="cuda")
    d = t.desc(L.GEMM_F16X2, torch.float16, torch.float32)
    ops.tdnn_split(xd, None, d, w, w_lo, None, None, None, y)
    got = y.cpu().numpy().astype(np.float64)
    x64 = x.astype(np.float64)
    Wh = W.astype(np.float16).astype(np.float64); Wl = (W.astype(np.float64) - Wh).astype(np.float16).astype(np.float64)
    full = x64 @ (Wh + Wl).T; hi_only = x64 @ Wh.T
    sub = np.mean(np.abs(Wl[Wl != 0]) < 6.1e-5)
    print(f"sigma {sigma}: |got - (hi+lo)| {np.abs(got-full).max():.3e}   |got - hi only| {np.abs(got-hi_only).max():.3e}   lo subnormal fraction {sub:.2f}")
