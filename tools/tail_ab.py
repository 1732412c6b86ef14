"""A/B of the x-vector tail at the bench batch (fused one-launch tail vs finalize + dense GEMM + post): ms per step of the whole
extractor and the kernels each variant launches after the pooled layer."""
import sys, time
sys.path[:0] = [".", "kaldi-tflite_amd", "tests"]
import torch, synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator(device="cuda").manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((B, 160000), generator=g, device="cuda")), -32767, 32767)
m = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f16mx")
ref = None
for fuse in (True, False, True, False):
    m.fuse_tail = fuse
    for _ in range(3): y = m(wav)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = m(wav)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    if ref is None: ref = y.clone()
    print(f"B {B} fuse_tail {fuse}: {dt*1e3:.3f} ms/step, last kernel {ops.last_kernel()}, max |diff| vs fused {float((y - ref).abs().max()):.2e}")
