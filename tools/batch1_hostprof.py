#!/usr/bin/env python3
"""Where the HOST time of a batch-1 extraction goes (cProfile over eager calls): python tools/batch1_hostprof.py [mode]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import synth
import kaldi_tflite_amd as ktf
mode = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
m = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm=mode)
wav = torch.as_tensor(synth.make_wav(1, 160000, seed=3), device="cuda")
for _ in range(50):
    m(wav)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    m(wav)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
