#!/usr/bin/env python3
"""ktf_xvec_post_f32 (mean subtraction, LDA, length normalisation) of the library in use on fixed inputs: results saved as .npy (compare two
builds bit by bit with KTF_LIBRARY + KTF_ALLOW_LIBRARY_OVERRIDE=1), and the launch timed at B = 1024 and 256.
  python tools/post_ab.py gpurun_out/post_new.npy"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from kaldi_tflite_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(9)
res = []
for B, din, dout in ((1024, 512, 150), (1027, 512, 150), (1, 512, 150), (300, 512, 128), (513, 200, 200), (258, 512, 300)):
    x = torch.randn((B, din), generator=g, device="cuda")
    mean = torch.randn((din,), generator=g, device="cuda") * 0.1
    A = torch.randn((din, dout), generator=g, device="cuda") / din ** 0.5
    off = torch.randn((dout,), generator=g, device="cuda") * 0.01
    y = ops.xvec_post(x, mean, A, off)
    res.append(y.cpu().numpy().ravel())
    if B in (1024, 300, 1):
        ms = bench._time_ms(torch, lambda: ops.xvec_post(x, mean, A, off, out=y), 50)
        print(f"{os.path.basename(_lib.LIB_PATH)}  B {B} {din} -> {dout}: {ms * 1e3:.1f} us per launch")
np.save(sys.argv[1], np.concatenate(res))
