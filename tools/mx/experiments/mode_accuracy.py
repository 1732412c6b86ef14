#!/usr/bin/env python3
"""max-abs x-vector deviation of the GEMM modes from the fp64 NumPy oracle at the full 10 s size, over several weight seeds
(checker run on the GPU box; uses oracle/ as tests do)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O
ktf.models.Sequential.MIN_TILES, ktf.models.Sequential.MIN_FRAMES = {}, {}
cfg = synth.extractor_cfg()
# "f16x2" = every layer two passes; "f16x2+cal" = calibrated: the two layers in front of the pooling run one pass (the default of bench.py)
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16x2+cal", "f16x2", "bf16x3", "f32"]
for seed in (4321, 1, 2, 3, 4, 5, 6, 7):
    w = synth.make_weights(seed=seed)
    N = 160000
    wav = np.concatenate([synth.make_wav(1, N, seed=1234), synth.make_wav(3, N, seed=4242 + seed, ragged=True)], 0)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    for g in modes:
        m = synth.build_extractor(ktf, cfg, w, gemm=g.split("+")[0], calibrate=g.endswith("+cal"))
        got = m(torch.as_tensor(wav, device="cuda")).cpu().numpy()
        print(f"weights seed {seed}  {g:9s} max-abs dev {np.abs(got - want).max():.3e}", flush=True)
