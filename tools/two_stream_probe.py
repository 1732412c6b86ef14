#!/usr/bin/env python3
"""Does running the TDNN stack of two HALF batches on two streams (out of phase by the host's enqueue time) hide the partial last round
of every GEMM launch (7984 workgroups = 31.2 rounds of 256 CUs per plane layer)? f16mx stack on CMVN-like features, 1024 x 998 frames:
one call on the whole batch against two concurrent calls on its halves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import kaldi_tflite_amd as ktf
dev = torch.device("cuda")
w = synth.make_weights(seed=4321)
seq = synth.build_sequential(ktf, w, "f16mx")
B, T = 1024, 998
x = torch.randn((B, T, 32), device=dev)[:, :, :30] * 0.5
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def whole():
    seq.run_ragged(x, lens)


def halves():
    cur = torch.cuda.current_stream()
    for st, sl in ((s1, slice(0, B // 2)), (s2, slice(B // 2, B))):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            seq.run_ragged(x[sl], lens[sl])
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def time_ms(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(3):
    print(f"whole batch {time_ms(whole):.3f} ms   two halves on two streams {time_ms(halves):.3f} ms")
