#!/usr/bin/env python3
"""The fused front-end launch alone (1024 x 998 frames of 400 samples -> 30 MFCCs), ms per launch, for the library in use
(KTF_LIBRARY + KTF_ALLOW_LIBRARY_OVERRIDE=1 select another build: same-box A/B of csrc/frontend512.hip variants)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import bench
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, int(os.environ.get("N_SAMPLES", 160000))), generator=g, device="cuda")), -32767, 32767)
for dither in (0.0, 1.0):
    m = synth.build_extractor(ktf, synth.extractor_cfg(dither=dither), synth.make_weights(seed=4321), gemm="f16mx")
    m(wav[:8])
    r = [bench._bench_mfcc(torch, m, wav, ops)["ms"] for _ in range(3)]
    print(os.path.basename(_lib.LIB_PATH), "dither", dither, "front-end ms", [round(x, 4) for x in r])
