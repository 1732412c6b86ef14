"""Runs the bench workload on an instrumented build (tools/mx/prof_stamps.py) and prints the summed phase times per (output kind, super-steps):
   KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=$PWD/_ab/libktf_prof.so python tools/mx/prof_run.py"""
import sys
sys.path[:0] = [".", "kaldi-tflite_amd", "tests"]
import torch, synth, bench
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device="cuda")), -32767, 32767)
m = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f16mx")
for _ in range(5): m(wav)
lib = _lib.load()
lib.ktf_prof_dump()
prof = bench._GemmProfiler(ops, torch)
n = 5
for _ in range(n): m(wav)
torch.cuda.synchronize()
st = prof.finish()
print(_lib.LIB_PATH, {k: round(v, 3) for k, v in st["per_layer_ms"].items()}, f"(per step; the sums below are over {n} steps)")
sys.stdout.flush()
lib.ktf_prof_dump()
