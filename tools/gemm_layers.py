#!/usr/bin/env python3
"""Times each frame-level TDNN GEMM of the 0008 topology at B=1024 x 998 frames (bf16) in isolation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L
B, T = int(os.environ.get("B", 1024)), 998
dev = torch.device("cuda", 0)
ONLY = os.environ.get("LAYERS", "tdnn2,tdnn4,tdnn5").split(",")
ITERS = int(os.environ.get("ITERS", 5))
for name, din, units, ctx in [("tdnn2", 512, 512, [-2, 0, 2]), ("tdnn4", 512, 512, [0]), ("tdnn5", 512, 1500, [0])]:
    if name not in ONLY:
        continue
    t = ktf.layers.TDNN(units, context=ctx, gemm="bf16")
    t.build((B, T, din))
    pad = int(os.environ.get("LDPAD", 0))      # extra columns in the row stride of x and y (L2 channel skew experiments)
    x = torch.randn((B, T, din + pad), device=dev).to(torch.bfloat16)
    y = torch.zeros((B, T, (units + 31) // 32 * 32 + pad), dtype=torch.bfloat16, device=dev)
    f = lambda: t.forward(x, relu=True, bn=None, gemm=L.GEMM_BF16, out_dtype=torch.bfloat16, ldy=y.shape[-1], out=y)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / ITERS
    fl = 2.0 * B * T * din * len(ctx) * units
    print(f"{name}: {ms:.3f} ms  {fl/ms/1e9:.0f} TF/s")
