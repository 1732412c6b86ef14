#!/usr/bin/env python3
"""Batch-1 (10 s utterance) extractor calls for a rocprofv3 --kernel-trace --stats run: which kernels make the latency."""
import os, sys
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
for _ in range(int(os.environ.get("N", "50"))):
    mdl(wav)
torch.cuda.synchronize()
