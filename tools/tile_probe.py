#!/usr/bin/env python3
"""Per-tile phase timestamps of the 128x256 bf16 TDNN kernel (needs a build with -DKTF_TILE_PROBE: make CXXFLAGS+=...).

The kernel then stores s_memrealtime stamps (setup, first stage landed, K-loop end, epilogue phases) per tile into the
buffer handed to the probe build's ktf_probe_set_buffer(); this script prints the mean phase times and the per-CU overlap.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, numpy as np
dev = torch.device("cuda", 0)
dbg = torch.zeros((65536, 16), dtype=torch.int64, device=dev)
os.environ.setdefault("KTF_HTILE", "1")
import ctypes
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L
L.load().ktf_probe_set_buffer(ctypes.c_void_p(dbg.data_ptr()))      # exists in -DKTF_TILE_PROBE builds only
B, T = 1024, 998
from kaldi_tflite_amd import ops
for name, din, units, ctx in [("tdnn2", 512, 512, [-2, 0, 2]), ("tdnn4", 512, 512, [0]), ("tdnn5", 512, 1500, [0]),
                              ("tdnn5+stats", 512, 1500, [0])]:
    t = ktf.layers.TDNN(units, context=ctx, gemm="bf16")
    t.build((B, T, din))
    x = torch.randn((B, T, din), device=dev).to(torch.bfloat16)
    y = torch.zeros((B, T, (units + 31) // 32 * 32), dtype=torch.bfloat16, device=dev)
    f = lambda: t.forward(x, relu=True, bn=None, gemm=L.GEMM_BF16, out_dtype=torch.bfloat16, ldy=y.shape[-1], out=y)
    if name.endswith("+stats"):
        sums = torch.zeros((B, 2, units), dtype=torch.float64, device=dev)
        w, w_lo, bias = t.device_weights(dev, L.GEMM_BF16)
        dsc = t.desc(L.GEMM_BF16, torch.bfloat16, torch.bfloat16, act="relu")
        f = lambda: ops.tdnn_stats(x, None, dsc, w, w_lo, bias, None, None, sums)
    f(); f(); torch.cuda.synchronize()
    dbg.zero_(); torch.cuda.synchronize()
    f(); torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    d = d[d[:, 0] != 0]
    t0, t1, t2, t3, t4 = (d[:, k].astype(np.float64) * 0.01 for k in range(5))   # 100 MHz -> us
    print(f"{name}: tiles {len(d)} nk {d[0,7]}  kernel span {t4.max()-t0.min():.1f} us")
    print(f"  setup {np.mean(t1-t0):.2f}  first-stage wait {np.mean(t2-t1):.2f}  loop {np.mean(t3-t2):.2f}  epilogue {np.mean(t4-t3):.2f}  total {np.mean(t4-t0):.2f} us")
    e = [d[:, k].astype(np.float64) * 0.01 for k in (8, 9, 10, 11)]
    print(f"  epi: sync-before {np.mean(e[0]-t3):.2f}  valu+ds_write {np.mean(e[1]-e[0]):.2f}  barrier {np.mean(e[2]-e[1]):.2f}  read+store {np.mean(t4-e[2]):.2f}")
    cyc = (d[:, 13] - d[:, 12]).astype(np.float64) / d[:, 7]
    print(f"  K-loop: {np.mean(cyc):.0f} shader cycles per K-step ({np.mean(t3-t2)/d[0,7]*1000:.0f} ns -> {np.mean(cyc)/(np.mean(t3-t2)/d[0,7]*1000):.2f} GHz); ideal 1024 with two workgroups per CU")
    # per-CU gaps
    key = (d[:, 6] & 0xf) * 65536 + (d[:, 5] & 0xff00)       # xcc, (se, sh, cu)
    gaps, busy = [], []
    for k in np.unique(key):
        m = key == k
        s = np.argsort(t0[m]); a0, a4 = t0[m][s], t4[m][s]
        gaps += list(a0[1:] - a4[:-1]); busy.append(len(s))
    gaps = np.array(gaps)
    print(f"  CUs {len(np.unique(key))} tiles/CU min {min(busy)} max {max(busy)}  inter-tile gap mean {gaps.mean():.2f} us  p50 {np.median(gaps):.2f}  p95 {np.percentile(gaps,95):.2f}")
    first = np.array([t0[key == k].min() for k in np.unique(key)]); last = np.array([t4[key == k].max() for k in np.unique(key)])
    print(f"  first-start spread {first.max()-first.min():.1f} us, last-end spread {last.max()-last.min():.1f} us")
