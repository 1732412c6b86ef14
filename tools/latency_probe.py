#!/usr/bin/env python3
"""Batch-1 latency of the 0008 extractor (BASELINE config 2): eager launches vs one captured HIP graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
import kaldi_tflite_amd as ktf

dev = torch.device("cuda", 0)
cfg = synth.extractor_cfg()
w = synth.make_weights()
for gemm in ("f32", "bf16x3", "bf16"):
    mdl = synth.build_extractor(ktf, cfg, w, gemm=gemm)
    wav = torch.as_tensor(synth.make_wav(1, 160000, seed=3), device=dev)
    ref = mdl(wav).clone()
    torch.cuda.synchronize()

    def timeit(fn, n=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    eager = timeit(lambda: mdl(wav))
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                mdl(wav)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = mdl(wav)
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        graph = timeit(g.replay)
        print(f"{gemm}: eager {eager:.3f} ms   graph replay {graph:.3f} ms   identical {same}")
    except Exception as e:  # noqa: BLE001
        print(f"{gemm}: eager {eager:.3f} ms   graph capture failed: {type(e).__name__}: {e}")
