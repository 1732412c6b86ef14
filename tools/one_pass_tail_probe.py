#!/usr/bin/env python3
"""max-abs x-vector deviation (x 1e-5, fp64 oracle, four 10 s utterances) of the calibrated f16x2 model for the settings of
Sequential.one_pass_tail x Sequential.lo_fraction, over weight seeds -- the real kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O
ktf.models.Sequential.min_tiles = {}
cfg = synth.extractor_cfg()
grid = [tuple(float(v) if '.' in v else int(v) for v in g.split(':')) for g in os.environ.get('GRID', '2:0.0,2:0.5,2:0.625,2:0.75,3:0.5,3:0.75').split(',')]
print("settings (one_pass_tail, lo_fraction):", grid)
worst = np.zeros(len(grid))
for seed in [4321] + list(range(1, int(os.environ.get('SEEDS', '8')))):
    w = synth.make_weights(seed=seed)
    N = 160000
    wav = np.concatenate([synth.make_wav(1, N, seed=1234), synth.make_wav(3, N, seed=4242 + seed, ragged=True)], 0)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    out = []
    for tail, frac in grid:
        ktf.models.Sequential.one_pass_tail, ktf.models.Sequential.lo_fraction = tail, frac
        m = synth.build_extractor(ktf, cfg, w, gemm="f16x2", calibrate=True)
        got = m(torch.as_tensor(wav, device="cuda")).cpu().numpy()
        out.append(np.abs(got - want).max())
    worst = np.maximum(worst, out)
    print(f"seed {seed}:", " ".join(f"{e*1e5:5.2f}" for e in out), flush=True)
print("worst    :", " ".join(f"{e*1e5:5.2f}" for e in worst))
