#!/usr/bin/env python3
"""Per-tile phase stamps of the split-plane kernel in the f16x2 step (probe build: make -C kaldi-tflite_amd/csrc probe;
KTF_LIBRARY=.../libktf_probe.so). Prints mean phase times per layer shape."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L, ops
dev = torch.device("cuda", 0)
dbg = torch.zeros((1 << 16, 8), dtype=torch.int64, device=dev)
L.load().ktf_probe_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm=os.environ.get("GEMM", "f16x2"))
g = torch.Generator(device=dev).manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device=dev)), -32767, 32767)
for _ in range(3): mdl(wav)
orig = {n: getattr(ops, n) for n in ("tdnn_split", "tdnn_split_stats")}
def wrap(name):
    def f(x, lens, desc, *a, **k):
        torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
        r = orig[name](x, lens, desc, *a, **k)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy(); d = d[d[:, 0] != 0]
        t = [d[:, k].astype(np.float64) * 0.01 for k in range(5)]
        print(f"{int(desc.nctx)}x{int(desc.din)}->{int(desc.units)}{'+stats' if 'stats' in name else ''}: tiles {len(d)}  span {t[4].max()-t[0].min():.0f} us | "
              f"setup+issue {np.mean(t[1]-t[0]):.2f}  first stage lands {np.mean(t[2]-t[1]):.2f}  K-loop {np.mean(t[3]-t[2]):.2f}  epilogue {np.mean(t[4]-t[3]):.2f}  tile {np.mean(t[4]-t[0]):.2f} us")
        return r
    return f
for n in orig: setattr(ops, n, wrap(n))
mdl(wav)
