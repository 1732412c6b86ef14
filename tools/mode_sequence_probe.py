"""Debug probe: per-call step time of the GEMM modes run one after another in one process (as bench.py's side
measurements do), optionally with the main model kept alive / after the bench helpers ran."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "kaldi-tflite_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import kaldi_tflite_amd as ktf, synth
import bench
dev = torch.device("cuda:0")
cfg = synth.extractor_cfg(); w = synth.make_weights(seed=4321, narrow=False)
g = torch.Generator(device=dev); g.manual_seed(1234)
wav = (torch.randn((1024, 160000), generator=g, device=dev) * 1000).round().clamp(-32767, 32767)
def t(fn, n):
    fn(); torch.cuda.synchronize(); out = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) * 1e3, 2))
    return out
steps = sys.argv[1].split(",")
keep = []
for s in steps:
    if s == "mfcc":
        m = synth.build_extractor(ktf, cfg, w, gemm="bf16"); m(wav); print("mfcc", bench._bench_mfcc(m, wav, ktf.ops)["ms"]); keep.append(m); continue
    if s.startswith("keep:"):
        m = synth.build_extractor(ktf, cfg, w, gemm=s[5:]); print(s, t(lambda: m(wav), 3)); keep.append(m); continue
    m = synth.build_extractor(ktf, cfg, w, gemm=s)
    print(s, t(lambda: m(wav), 4), "ms", round(torch.cuda.memory_allocated() / 2**30, 1), "GiB alloc")
    del m; torch.cuda.empty_cache()
