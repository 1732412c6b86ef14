import os, sys
sys.path.insert(0, "/root/repo/kaldi-tflite_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch, kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L, ops
import synth
dev = torch.device("cuda", 0)
B, T, D = 1024, 998, 30
mfcc = torch.randn((B, T, D), device=dev) * 3
mfcc[:, :, 0] = 20 + torch.randn((B, T), device=dev)
feats = torch.zeros((B, T, 32), dtype=torch.bfloat16, device=dev)
lens = torch.zeros((B,), dtype=torch.int32, device=dev); idx = torch.empty((B, T), dtype=torch.int32, device=dev)
work = torch.empty((B*T*2*D + 2*D,), device=dev)
vad = ktf.layers.VAD(energy_mean_scale=0.5, energy_threshold=5.5, frames_context=2, proportion_threshold=0.12)
def run(win):
    c = ktf.layers.CMVN(window=win)
    f = lambda: ops.vad_cmvn(mfcc, vad.cfg(), c.cfg(), feats, lens, idx, work)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    print("window", win, s.elapsed_time(e)/5, "ms  lens", int(lens.min()), int(lens.max()))
run(300); run(2000); run(100); run(600)
i2, l2 = ops.vad_index(mfcc, vad.cfg())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): ops.vad_index(mfcc, vad.cfg())
e.record(); torch.cuda.synchronize(); print("vad_index only", s.elapsed_time(e)/5)
