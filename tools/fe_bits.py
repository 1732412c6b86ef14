#!/usr/bin/env python3
"""MFCCs of one fixed batch (16 x 3 s: noise, a quiet and an all-zero utterance) from the library in use, saved as .npy next to the fp64
oracle's for the first six utterances -- for comparing two builds of csrc/frontend512.hip bit by bit and against the oracle:
  KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=<old .so> python tools/fe_bits.py gpurun_out/old.npy;  python tools/fe_bits.py gpurun_out/new.npy
Test infrastructure: the oracle is the checker."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch, numpy as np, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L, ops
g = torch.Generator(device="cuda").manual_seed(5)
wav = torch.clamp(torch.round(3000.0 * torch.randn((16, 48000), generator=g, device="cuda")), -32767, 32767)
wav[3] *= 1e-3; wav[4] = 0
m = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f32")
m(wav[:2])
fr, mf = m.framing, m.mfcc
B, N = wav.shape
T = fr.numFrames(N)
cfg = L.FrontendCfg.from_buffer_copy(mf._cfg)
cfg.frame_size, cfg.frame_shift = fr.frameWidth, fr.frameShift
out = torch.empty((B, T, mf.numMfccs), dtype=torch.float32, device=wav.device)
ops.frontend(wav, L.IN_WAV, cfg, mf.tables(wav.device), L.OUT_MFCC, N, B, T, out=out)
torch.cuda.synchronize()
np.save(sys.argv[1], out.cpu().numpy())
print(os.path.basename(L.LIB_PATH), out.shape)

# fp64 oracle of the same frames (test infrastructure)
from oracle import ktf_oracle as O
c = synth.extractor_cfg()
w = wav[:6].cpu().numpy().astype(np.float64)
fk = {k: v for k, v in c["framing"].items() if k != "dynamic_input_shape"}
fo = O.mfcc(O.framing(w, **fk), dtype=np.float64, **c["mfcc"])
np.save(sys.argv[1].replace(".npy", "_oracle.npy"), fo)
