#!/usr/bin/env python3
"""x-vectors of one fixed batch from the library in use, saved as .npy -- for comparing two builds of the f16mx kernels bit by bit:
  KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=<old .so> python tools/xvec_bits.py gpurun_out/old.npy;  python tools/xvec_bits.py gpurun_out/new.npy
The batch: 96 utterances of 10 s of noise whose voiced lengths differ (quiet blocks of 0 .. 3 s at the front, in the middle or at the end:
pooling runs that start and end anywhere inside the flat 128-row blocks), one all-quiet utterance, one of 2 s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch, numpy as np, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L
g = torch.Generator(device="cuda").manual_seed(77)
B, N = 96, 160000
wav = torch.clamp(torch.round(3000.0 * torch.randn((B, N), generator=g, device="cuda")), -32767, 32767)
rs = np.random.RandomState(3)
for b in range(B):
    q = int(rs.randint(0, 48000))
    at = [0, (N - q) // 2, N - q][b % 3]
    wav[b, at:at + q] *= 1e-4
wav[5] *= 1e-4
wav[7, 32000:] *= 1e-4
out = {}
for gemm in ("f16mx", "bf16x3"):
    m = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm=gemm)
    y = m(wav)
    torch.cuda.synchronize()
    out[gemm] = y.cpu().numpy()
np.save(sys.argv[1], np.stack([out["f16mx"], out["bf16x3"]]))
print(os.path.basename(L.LIB_PATH), out["f16mx"].shape, float(np.nanmax(np.abs(out["f16mx"] - out["bf16x3"]))))
