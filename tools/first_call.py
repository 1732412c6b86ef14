#!/usr/bin/env python3
"""Cold-start cost per mode: model construction, the first call (weight images built on the host and uploaded, tables, LDS opt-ins),
the second call: python tools/first_call.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import kaldi_tflite_amd as ktf
torch.zeros(1, device="cuda")
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
for B in (1, 64):
    wav = torch.as_tensor(synth.make_wav(B, 160000, seed=3), device="cuda")
    for gemm in ("f32", "bf16x3", "f16mx"):
        t0 = time.perf_counter(); m = synth.build_extractor(ktf, cfg, w, gemm=gemm); t1 = time.perf_counter()
        m(wav); torch.cuda.synchronize(); t2 = time.perf_counter()
        m(wav); torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"B={B:3d} {gemm:7s} build {t1 - t0:.3f} s  first call {t2 - t1:.3f} s  second call {(t3 - t2) * 1e3:.2f} ms")
