#!/usr/bin/env python3
"""Does the VALU-bound front-end of one half batch run beside the MFMA-bound TDNN GEMMs of the other (two streams)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, synth
import kaldi_tflite_amd as ktf
dev = torch.device("cuda", 0)
mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f16mx")
g = torch.Generator(device=dev).manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device=dev)), -32767, 32767)
wa, wb = wav[:512], wav[512:]
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def feats(w):
    _, f, l = mdl._features(w)
    return f, l

fa, la = feats(wa)
fa, la = fa.clone(), la.clone()
t_fe = timeit(lambda: feats(wb))
t_gemm = timeit(lambda: mdl.xvec.run_ragged(fa, la))


def both():
    side.wait_stream(main)
    mdl.xvec.run_ragged(fa, la)            # GEMMs first: one workgroup per CU, the front-end's workgroups fit beside them
    with torch.cuda.stream(side):
        feats(wb)
    main.wait_stream(side)

t_both = timeit(both)
print(f"{os.path.basename(ktf._lib.LIB_PATH)}: features(512) {t_fe:.3f} ms, TDNN stack(512) {t_gemm:.3f} ms, sum {t_fe + t_gemm:.3f}, on two streams {t_both:.3f} ms")
