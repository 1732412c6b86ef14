"""Per-tile phase times of the loader-wave f16mx kernel from the instrumented build (tools/mx/ablate_mxl.py l_prof):
   KTF_ALLOW_LIBRARY_OVERRIDE=1 KTF_LIBRARY=$PWD/kaldi-tflite_amd/kaldi_tflite_amd/libktf_abl_l_prof.so python tools/mx/prof_mxl.py
Shader-clock cycles per tile: the loader wave's prologue (until the first stage has landed), then over the K-loop its issue time,
its waits for landed DMAs and its barrier waits; matrix wave 0's K-loop (of which: barrier waits) and epilogue."""
import sys
sys.path[:0] = [".", "kaldi-tflite_amd", "tests"]
import torch, synth, bench
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device="cuda")), -32767, 32767)
m = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f16mx")
m.xvec.mx_loader = True          # (the 256-row kernel is the default)
for _ in range(3): m(wav)
lib = _lib.load()
print("library:", _lib.LIB_PATH)
lib.ktf_xprof_dump()
prof = bench._GemmProfiler(ops, torch)
n = 5
for _ in range(n): m(wav)
torch.cuda.synchronize()
st = prof.finish()
print({k: round(v, 3) for k, v in st["per_layer_ms"].items()}, f"(per step; the sums below are over {n} steps)")
sys.stdout.flush()
lib.ktf_xprof_dump()
