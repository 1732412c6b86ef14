#!/usr/bin/env python3
"""Single-utterance latency (10 s, config 2) of one library build: eager and hipGraph replay, median over repeats.
usage: [KTF_LIBRARY=...] b1_time.py [gemm]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import synth
import kaldi_tflite_amd as ktf

gemm = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda", 0)
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(), gemm=gemm)
wav = torch.as_tensor(synth.make_wav(1, 160000, seed=3), device=dev)


def med(fn, n=300, reps=7):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / n * 1e3)
    return sorted(out)[len(out) // 2]


for _ in range(50):
    y = mdl(wav)
eager = med(lambda: mdl(wav))
g_ = mdl.compile(wav)
g = lambda: g_(wav)
for _ in range(20):
    g()
graph = med(g)
print(f"{os.path.basename(ktf._lib.LIB_PATH):>60s}  eager {eager:.4f} ms   graph {graph:.4f} ms   checksum {float(y.double().sum()):.9f}")
