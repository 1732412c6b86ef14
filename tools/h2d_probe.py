"""Host->device upload probe: raw pinned-memory copy rate, and extract_stream with the copy hidden under compute."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-tflite_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import kaldi_tflite_amd as ktf
import synth

dev = torch.device("cuda:0")
B, N = 1024, 160000
host16 = torch.randint(-3000, 3000, (B, N), dtype=torch.int16).pin_memory()
d = torch.empty((B, N), dtype=torch.int16, device=dev)
for _ in range(2):
    d.copy_(host16, non_blocking=True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    d.copy_(host16, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"raw H2D int16 batch: {dt*1e3:.2f} ms = {host16.numel()*2/dt/1e9:.1f} GB/s")

cfg = synth.extractor_cfg()
w = synth.make_weights(seed=4321, narrow=False)
mdl = synth.build_extractor(ktf, cfg, w, gemm="bf16")
mdl(d); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(4):
    mdl(d)
torch.cuda.synchronize()
print(f"compute only: {(time.perf_counter()-t)/4*1e3:.2f} ms/batch")
hosts = [host16, host16.clone().pin_memory(), host16.clone().pin_memory()]
for depth in (2, 3):
    for nb in (8,):
        outs = list(mdl.extract_stream([hosts[i % 3] for i in range(2)], depth=depth)); torch.cuda.synchronize()
        t = time.perf_counter()
        th = []
        for y in mdl.extract_stream([hosts[i % 3] for i in range(nb)], depth=depth):
            th.append(time.perf_counter() - t)
        t_host = time.perf_counter() - t
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"depth {depth}: {dt/nb*1e3:.2f} ms/batch = {B*nb/dt:.0f} utt/s; host enqueue done at {t_host*1e3:.1f} ms; yields at {[round(x*1e3,1) for x in th]}")
