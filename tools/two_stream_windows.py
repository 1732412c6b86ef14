#!/usr/bin/env python3
"""1024 windows of 1.5 s through the shipped f16mx extractor (short-utterance route: split-bf16 on flat row tiles): one call on the whole
batch against two concurrent calls on its halves on two streams. The full-size layers launch 1184 workgroups = 4.6 rounds of 256 CUs, so
the partial last round of every launch is 7.5 % of it; two streams out of phase let one half's tiles fill the other's last round."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import kaldi_tflite_amd as ktf
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
g = torch.Generator(device="cuda").manual_seed(77)
B = int(os.environ.get("B", "1024"))
wav = torch.clamp(torch.round(1000.0 * torch.randn((B, 24000), generator=g, device="cuda")), -32767, 32767)
m = synth.build_extractor(ktf, cfg, w, gemm=os.environ.get("GEMM", "f16mx"))
NS = int(os.environ.get("NS", "2"))
streams = [torch.cuda.Stream() for _ in range(NS)]
out = {}


def whole():
    out["w"] = m(wav)


def parts():
    cur = torch.cuda.current_stream()
    n = B // NS
    for i, st in enumerate(streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            out[i] = m(wav[i * n:(i + 1) * n])
    for st in streams:
        cur.wait_stream(st)


def time_ms(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(3):
    a, b = time_ms(whole), time_ms(parts)
    print(f"whole batch {a:.3f} ms ({B / a * 1e3:.0f} windows/s)   {NS} parts on {NS} streams {b:.3f} ms ({B / b * 1e3:.0f} windows/s)")
whole(); parts(); torch.cuda.synchronize()
got = torch.cat([out[i] for i in range(NS)])
print("parts == whole bit for bit:", bool(torch.equal(got, out["w"])), " max diff", float((got - out["w"]).abs().max()))
