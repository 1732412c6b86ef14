#!/usr/bin/env python3
"""Throughput on SHORT utterances (diarization-sized windows: SURVEY 8 f4): 1024 windows of 1.5 s / 3 s per step, ms per step for the
f16mx model as shipped (utterances under 400 frames -> the split-bf16 256-row kernels), on the bf16-pair small tiles, and in the other
modes: python tools/short_windows.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import bench
import kaldi_tflite_amd as ktf
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
g = torch.Generator(device="cuda").manual_seed(1234)
for sec in (1.5, 3.0):
    n = int(16000 * sec)
    wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, n), generator=g, device="cuda")), -32767, 32767)
    row = {}
    for name, gemm, tiles in (("f16mx as shipped (-> bf16x3 256-row)", "f16mx", None), ("bf16-pair small tiles", "f16mx", 10 ** 9),
                              ("bf16x3", "bf16x3", None), ("f32", "f32", None), ("f16mx kernels (outside the tolerance here)", "f16mx", -1)):
        m = synth.build_extractor(ktf, cfg, w, gemm=gemm)
        if tiles == -1:
            m.xvec.min_frames = {}
        elif tiles is not None:
            m.xvec.min_tiles = {gemm: tiles, "bf16x3": tiles}
        row[name] = bench._time_ms(torch, lambda: m(wav), 10)
    print(f"{sec} s x 1024:", {k: round(v, 3) for k, v in row.items()})
