#!/usr/bin/env python3
"""ms per step of small batches of 10 s utterances on the small-tile route (bf16 pairs / fp32) and on the mode's own 256-row kernels:
where Sequential.MIN_TILES should put the crossover. python tools/small_batch_crossover.py [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import synth
import bench
import kaldi_tflite_amd as ktf
mode = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
wav = torch.as_tensor(synth.make_wav(64, 160000, seed=3), device="cuda")
print(f"mode {mode}: B | small tiles (pairs) | small tiles (fp32) | 256-row kernels   [ms per step]")
for B in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64):
    x = wav[:B].contiguous()
    row = []
    for tiles, pairs in ((10 ** 9, True), (10 ** 9, False), (0, True)):
        m = synth.build_extractor(ktf, cfg, w, gemm=mode)
        m.xvec.min_tiles = {mode: tiles}
        m.xvec.small_tile_pairs = pairs
        m.route_short_utterances = False
        run = m.compile(x)
        row.append(bench._time_ms(torch, lambda: run(x), 30))
    print(f"{B:3d} | {row[0]:.4f} | {row[1]:.4f} | {row[2]:.4f}")
