import sys, time
sys.path[:0]=[".", "kaldi-tflite_amd", "tests"]
import torch, numpy as np, synth, bench
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
g = torch.Generator(device="cuda").manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device="cuda")), -32767, 32767)
for name in ("random weights", "zero weights (zero operands from tdnn2 on)"):
    w = synth.make_weights(seed=4321)
    if name.startswith("zero"):
        for k in list(w):
            if k.endswith(".affine"):
                w[k] = (np.zeros_like(w[k][0]), np.zeros_like(w[k][1]))
    m = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm="f16mx")
    for _ in range(3): m(wav)
    prof = bench._GemmProfiler(ops, torch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(wav)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    st = prof.finish()
    print(name, f"{dt*1e3:.3f} ms/step", {k: round(v, 3) for k, v in st["per_layer_ms"].items()})
