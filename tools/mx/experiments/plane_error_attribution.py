#!/usr/bin/env python3
"""Which stored half plane makes the f16x2 base error? fp64 oracle (checker; CPU only) with the activations rounded to half
at chosen points: P0 = CMVN'd features (tdnn1's input), P1..P4 = ReLU outputs of tdnn1..tdnn4 (what the route stores;
BatchNorm applied afterwards in fp64 = folded forward). Weights exact."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, synth
from oracle import ktf_oracle as O

cfg = synth.extractor_cfg()
N = 160000


def forward(wav, layers, w, rounded):
    outs = []
    fcfg = {k: v for k, v in cfg["framing"].items() if k != "dynamic_input_shape"}
    for b in range(wav.shape[0]):
        fr = O.framing(wav[b:b + 1].astype(np.float64), **fcfg)
        m = O.mfcc(fr, **cfg["mfcc"], dtype=np.float64)
        vcfg = dict(cfg["vad"]); vcfg["return_indexes"] = True
        idx = O.vad(m, **vcfg, dtype=np.float64)
        x = O.cmvn(m[idx[:, 0], idx[:, 1]][None], **cfg["cmvn"], dtype=np.float64)
        if 0 in rounded:
            x = x.astype(np.float16).astype(np.float64)
        plane = 0
        for L in layers:
            k = L["kind"]
            if k == "tdnn":
                x = O.tdnn(x, L["W"], L.get("b"), L.get("context", [0]), dtype=np.float64)
            elif k == "relu":
                x = O.relu(x)
                plane += 1
                if plane in rounded:
                    x = x.astype(np.float16).astype(np.float64)
            elif k == "bn":
                x = O.batchnorm(x, L["rms"], L["mean"], L["var"], 1e-3, dtype=np.float64)
            elif k == "stats":
                x = O.stats_pooling(x, **{a: c for a, c in L.items() if a != "kind"}, dtype=np.float64)
        outs.append(O.xvector_post(x, w["mean"], w["lda"], dtype=np.float64)[0])
    return np.stack(outs, 0)


for seed in (4321, 1, 2):
    w = synth.make_weights(seed=seed)
    layers = synth.oracle_layers(w)
    wav = np.concatenate([synth.make_wav(1, N, seed=1234), synth.make_wav(3, N, seed=4242 + seed, ragged=True)], 0)
    exact = forward(wav, layers, w, ())
    res = []
    for r in ((0,), (1,), (2,), (3,), (4,), (0, 1, 2, 3, 4)):
        res.append(np.abs(forward(wav, layers, w, r) - exact).max())
    print(f"seed {seed}: max-abs x1e-5 with only P0 / P1 / P2 / P3 / P4 rounded, all:", " ".join(f"{e*1e5:5.2f}" for e in res), flush=True)
