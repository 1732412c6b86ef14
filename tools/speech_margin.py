#!/usr/bin/env python3
"""How much room the compliant GEMM modes leave under north_star's 1e-4: max-abs x-vector deviation from the fp64 oracle over
sliding windows of the reference's speech recording (tests/golden/e2e_0008.npz: 10 s and 5 s windows at 1 s strides, the whole
22.5 s), amplitude-modulated coloured noise and stationary noise, for several weight seeds. Checker run on the GPU box (uses
oracle/ as the tests do):   python tools/speech_margin.py [modes] [seeds]   e.g.  f16mx,bf16x3 4321,1,2,3,4,5,6,7"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O

ktf.models.Sequential.MIN_TILES, ktf.models.Sequential.MIN_FRAMES = {}, {}       # every batch runs the mode under test (by default a handful of tiles goes to the fp32 kernels)
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16mx"]
seeds = [int(s) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4321, 1, 2, 3]
cfg = synth.extractor_cfg()
whole, _ = synth.speech_wavs()
sp = whole[0]
sets = {}
n10, n5 = 160000, 80000
sets["speech 10 s windows"] = np.stack([sp[o:o + n10] for o in range(0, len(sp) - n10 + 1, 16000)], 0)
sets["speech 5 s windows"] = np.stack([sp[o:o + n5] for o in range(0, len(sp) - n5 + 1, 16000)], 0)
sets["speech 22.5 s"] = whole
sets["coloured AM noise 10 s"] = synth.coloured_am_noise(4, n10, seed=99)
sets["stationary noise 10 s"] = np.concatenate([synth.make_wav(2, n10, seed=1234), synth.make_wav(2, n10, seed=4242, ragged=True)], 0)
print({k: v.shape for k, v in sets.items()}, flush=True)
worst = {m: 0.0 for m in modes}
allv = {m: [] for m in modes}
for seed in seeds:
    w = synth.make_weights(seed=seed)
    layers = synth.oracle_layers(w)
    mdl = {m: synth.build_extractor(ktf, cfg, w, gemm=m) for m in modes}
    for name, wav in sets.items():
        want = O.xvector_forward(wav, cfg, layers, w["mean"], w["lda"], dtype=np.float64)
        line = f"weights seed {seed:5d}  {name:24s}"
        for m in modes:
            got = mdl[m](torch.as_tensor(wav, device="cuda")).cpu().numpy().reshape(want.shape)
            d = np.abs(got - want).max(-1)                       # per utterance
            allv[m] += list(d)
            worst[m] = max(worst[m], float(d.max()))
            line += f"  {m}: max {d.max():.2e} median {np.median(d):.2e}"
        print(line, flush=True)
for m in modes:
    v = np.sort(np.asarray(allv[m]))
    print(f"{m}: {len(v)} x-vectors, max {v[-1]:.2e}, p99 {v[int(0.99 * (len(v) - 1))]:.2e}, median {np.median(v):.2e}, "
          f"share above 5e-5: {float((v > 5e-5).mean()):.3f}, above 1e-4: {float((v > 1e-4).mean()):.3f}")
