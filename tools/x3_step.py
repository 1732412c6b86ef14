#!/usr/bin/env python3
"""Times the default (split-bf16) 1024 x 10 s step and its GEMM launches without any result checks (for timing-only ablation
builds selected with KTF_LIBRARY)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, synth, bench
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
dev = torch.device("cuda", 0)
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm=os.environ.get("GEMM", "bf16x3"))
g = torch.Generator(device=dev).manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device=dev)), -32767, 32767)
for _ in range(3): mdl(wav)
prof = bench._GemmProfiler(ops, torch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): mdl(wav)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
st = prof.finish()
print(os.path.basename(ktf._lib.LIB_PATH), f"{dt*1e3:.3f} ms/step", {k: round(v, 3) for k, v in st["per_layer_ms"].items()})
