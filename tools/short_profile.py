#!/usr/bin/env python3
"""The shipped f16mx model on 1024 windows of 1.5 s (148 frames: the short-utterance route, split-bf16 on flat row tiles), 30 steps --
for `rocprofv3 --kernel-trace --stats -- python3 tools/short_profile.py` (per-kernel times of the route)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import bench
import kaldi_tflite_amd as ktf
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
g = torch.Generator(device="cuda").manual_seed(77)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 24000), generator=g, device="cuda")), -32767, 32767)
m = synth.build_extractor(ktf, cfg, w, gemm=os.environ.get("GEMM", "f16mx"))
if os.environ.get("FLAT_MAP", "1") == "0":          # A/B: every workgroup derives its row table itself
    from kaldi_tflite_amd import ops
    _fr = ops.flat_rows
    ops.flat_rows = lambda lens, B, T, get: ops.FlatRows(_fr(lens, B, T, get).starts, None)
if os.environ.get("FLAT_POOLING", "1") == "0":      # A/B: the pooled layer on per-utterance tiles
    m.xvec.flat_pooling = False
ms = bench._time_ms(torch, lambda: m(wav), int(os.environ.get("STEPS", "30")))
print(f"1024 x 1.5 s: {ms:.3f} ms per step = {1024 / ms * 1e3:.0f} windows/s")
