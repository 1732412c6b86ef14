#!/usr/bin/env python3
"""Per-workgroup phase stamps of the single-utterance fp32 TDNN kernel (probe build: make -C kaldi-tflite_amd/csrc probe;
KTF_LIBRARY=.../libktf_probe.so): launch skew, time to the first stage, K-loop, epilogue, per layer of one 10 s utterance."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L, ops
dev = torch.device("cuda", 0)
dbg = torch.zeros((1 << 14, 8), dtype=torch.int64, device=dev)
L.load().ktf_probe_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f32")
wav = torch.as_tensor(synth.make_wav(1, 160000, seed=3), device=dev)
for _ in range(20): mdl(wav)
orig = ops.tdnn
def wrapped(x, lens, desc, *a, **k):
    for rep in range(3):
        torch.cuda.synchronize(); dbg.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(x, lens, desc, *a, **k); e1.record()
        torch.cuda.synchronize()
    d = dbg.cpu().numpy(); d = d[d[:, 0] != 0]
    if len(d):
        t = [d[:, k].astype(np.float64) * 0.01 for k in range(4)]
        clk = (d[:, 6] - d[:, 5]).astype(np.float64) / np.maximum((d[:, 2] - d[:, 1]).astype(np.float64) * 10.0, 1.0)     # shader cycles per ns over the K-loop
        print(f"shader clock over the K-loop: {np.median(clk):.2f} GHz", end="   ")
        print(f"{int(desc.nctx)}x{int(desc.din)}->{int(desc.units)}: workgroups {len(d)}  event {e0.elapsed_time(e1) * 1e3:.1f} us  span {t[3].max() - t[0].min():.1f} us | "
              f"start skew {t[0].max() - t[0].min():.2f}  first stage {np.mean(t[1] - t[0]):.2f}  K-loop {np.mean(t[2] - t[1]):.2f} (max {np.max(t[2] - t[1]):.2f})  epilogue {np.mean(t[3] - t[2]):.2f}  end skew {t[3].max() - t[3].min():.2f}")
    return r
ops.tdnn = wrapped
mdl(wav)
