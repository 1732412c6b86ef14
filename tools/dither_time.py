#!/usr/bin/env python3
"""The 1024 x 10 s step and its feature stage (front-end + VAD / CMVN) with the MFCC dither off and at the reference's shipped default
(data/tflite_models/0008_sitw_v2_1a.yml: dither 1.0): python tools/dither_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch, synth, bench
import kaldi_tflite_amd as ktf
w=synth.make_weights(seed=4321)
g=torch.Generator(device="cuda").manual_seed(1234)
wav=torch.clamp(torch.round(1000.0*torch.randn((1024,160000),generator=g,device="cuda")),-32767,32767)
for d in (0.0,1.0):
    m=synth.build_extractor(ktf, synth.extractor_cfg(dither=d), w, gemm="f16mx")
    ms=bench._time_ms(torch, lambda: m(wav), 10)
    fe=bench._time_ms(torch, lambda: m.features(wav), 10)
    print("dither",d,"step ms",round(ms,3),"features (front-end + VAD/CMVN) ms",round(fe,3))
