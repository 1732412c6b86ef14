#!/usr/bin/env python3
"""Segment stamps (shader clock) of the ping-pong K-loop, K-step 10 of the first 4096 tiles of each split-plane launch
(probe build). Group 0: MFMA | barrier | DMA issue | fragment reads; group 1: DMA issue | reads | barrier | MFMA."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L, ops
dev = torch.device("cuda", 0)
dbg = torch.zeros((65536 + 8192, 8), dtype=torch.int64, device=dev)
L.load().ktf_probe_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f16x2")
g = torch.Generator(device=dev).manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device=dev)), -32767, 32767)
for _ in range(3): mdl(wav)
orig = {n: getattr(ops, n) for n in ("tdnn_split", "tdnn_split_stats")}
def wrap(name):
    def f(x, lens, desc, *a, **k):
        torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
        r = orig[name](x, lens, desc, *a, **k)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy()[65536:].reshape(-1, 2, 8).astype(np.float64)
        d = d[(d[:, 0, 0] != 0) & (d[:, 1, 0] != 0)]
        if len(d):
            g0, g1 = np.diff(d[:, 0, :5], axis=1), np.diff(d[:, 1, :5], axis=1)
            print(f"{int(desc.nctx)}x{int(desc.din)}->{int(desc.units)}: group 0 cycles  MFMA {np.median(g0[:,0]):.0f}  wait+barrier {np.median(g0[:,1]):.0f}  DMA issue {np.median(g0[:,2]):.0f}  reads {np.median(g0[:,3]):.0f}"
                  f" | group 1  DMA issue {np.median(g1[:,0]):.0f}  reads {np.median(g1[:,1]):.0f}  wait+barrier {np.median(g1[:,2]):.0f}  MFMA {np.median(g1[:,3]):.0f}")
        return r
    return f
for n in orig: setattr(ops, n, wrap(n))
mdl(wav)
