#!/usr/bin/env python3
"""Phase stamps of the fused VAD + CMVN kernel on one 10 s utterance (probe build: KTF_LIBRARY=.../libktf_probe.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L
dev = torch.device("cuda", 0)
dbg = torch.zeros((16,), dtype=torch.int64, device=dev)
lib = L.load()
lib.ktf_probe_set_vc_buffer.argtypes = [ctypes.c_void_p]
lib.ktf_probe_set_vc_buffer(ctypes.c_void_p(dbg.data_ptr()))
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f32")
wav = torch.as_tensor(synth.make_wav(1, 160000, seed=3), device=dev)
names = ["C0 mean", "vote + scan", "(sync)", "stage rows into LDS", "block sums", "windows + stores", "edge frames", "end"]
acc = np.zeros(7)
n = 0
for it in range(30):
    mdl(wav)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()[:8].astype(np.float64) * 0.01
    if it >= 10:
        acc += np.diff(d)
        n += 1
acc /= n
print("  ".join(f"{nm}: {v:.2f}" for nm, v in zip(["C0 mean", "vote+scan", "stage", "block sums", "windows+stores", "edges", "tail"], acc)), f"| total {acc.sum():.2f} us")
