#!/usr/bin/env python3
"""Reference point for the roofline: torch.matmul (hipBLASLt / rocBLAS) bf16 on the plain-GEMM shapes of the TDNN layers
(activations already spliced, i.e. without the implicit im2col the TDNN kernels do). Not part of the product."""
import torch
dev = torch.device("cuda", 0)
M = 1024 * 998
for name, K, N in [("tdnn2/3", 1536, 512), ("tdnn4", 512, 512), ("tdnn5", 512, 1500), ("square 8192", 8192, 8192)]:
    m = 8192 if name.startswith("square") else M
    a = torch.randn((m, K), device=dev).to(torch.bfloat16)
    w = torch.randn((N, K), device=dev).to(torch.bfloat16)
    f = lambda: torch.matmul(a, w.t())
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"{name}: M={m} K={K} N={N}: {ms:.3f} ms  {2.0*m*K*N/ms/1e9:.0f} TF/s")
