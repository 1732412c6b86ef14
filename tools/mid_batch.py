#!/usr/bin/env python3
"""f16mx on 5 ... 128 utterances of 10 s: ms per step (captured graph) on the 256-row kernel and the loader-wave kernel
-- where `Sequential._mx_use_loader` (mx_loader = None) switches: python tools/mid_batch.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, synth, bench
import kaldi_tflite_amd as ktf
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
wav = torch.as_tensor(synth.make_wav(128, 160000, seed=3), device="cuda")
print("B | 256-row | loader  [ms per step, f16mx forced]")
for B in (5, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128):
    x = wav[:B].contiguous()
    row = []
    for kind in ("tile", "loader"):
        m = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
        m.xvec.min_tiles = {}
        m.xvec.mx_loader = kind == "loader"
        run = m.compile(x)
        row.append(bench._time_ms(torch, lambda: run(x), 30))
    print(f"{B:3d} | {row[0]:.4f} | {row[1]:.4f}")
