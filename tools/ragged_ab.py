#!/usr/bin/env python3
"""f16mx plane layers on flat row tiles against per-utterance tiles (Sequential.mx_flat_rows) on batches the VAD leaves ragged: 1024 x 10 s
utterances with a fraction of their 0.5 s blocks quiet -- python tools/ragged_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import bench
import kaldi_tflite_amd as ktf
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device=dev)), -32767, 32767)
for frac in [float(v) for v in os.environ.get("FRACS", "0.0,0.1,0.3,0.5").split(",")]:
    quiet = torch.rand((1024, 20), device=dev, generator=torch.Generator(device=dev).manual_seed(5)) < frac
    quiet[:, 0] = False
    x = torch.round(wav * torch.where(quiet, 1e-3, 1.0).repeat_interleave(8000, dim=1))
    row = {}
    for rep in range(2):
        for flat in ((False, True) if os.environ.get("ORDER") == "rev" else (True, False)):
            m = synth.build_extractor(ktf, cfg, w, gemm=os.environ.get("GEMM", "f16mx"))
            m.xvec.mx_flat_rows = flat
            m.xvec.flat_rows_long = flat if os.environ.get("LONG_AB", "1") == "1" else True
            ms = bench._time_ms(torch, lambda: m(x), 10)
            row.setdefault(flat, []).append(ms)
            lens = m.last_lens.float()
    print(f"quiet fraction {frac}: mean voiced frames {float(lens.mean()):.0f} (min {int(lens.min())}); ms per step flat tiles {['%.3f' % v for v in row[True]]}, "
          f"per-utterance tiles {['%.3f' % v for v in row[False]]}")
