import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O
ktf.models.Sequential.min_tiles = {}
cfg = synth.extractor_cfg()
for seed in (4321, 1, 2, 3, 4, 5, 6, 7):
    w = synth.make_weights(seed=seed)
    N = 160000
    wav = np.concatenate([synth.make_wav(1, N, seed=1234), synth.make_wav(3, N, seed=4242 + seed, ragged=True)], 0)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    out = []
    for tail in (1, 2, 3, 4, 5):
        ktf.models.Sequential.one_pass_tail = tail
        m = synth.build_extractor(ktf, cfg, w, gemm="f16x2", calibrate=True)
        got = m(torch.as_tensor(wav, device="cuda")).cpu().numpy()
        out.append(np.abs(got - want).max())
    print(f"seed {seed}: one-pass tail 1..5:", " ".join(f"{e*1e5:5.2f}" for e in out), flush=True)
