#!/usr/bin/env python3
"""Per-step times of the default 1024 x 10 s step from a cold start: how long until the GPU reaches its sustained state?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, synth
import kaldi_tflite_amd as ktf
dev = torch.device("cuda", 0)
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm=os.environ.get("GEMM", "f16mx"))
g = torch.Generator(device=dev).manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device=dev)), -32767, 32767)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE", "2")))
ts = []
for i in range(int(os.environ.get("N", "80"))):
    t = time.perf_counter()
    mdl(wav)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
