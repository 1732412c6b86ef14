#!/usr/bin/env python3
"""Host threads calling concurrently: every x-vector must equal the single-threaded result bit for bit.
   python tools/thread_probe.py [calls per thread]                  each thread its own XvectorExtractor and its own HIP stream
                                                                    (per-model workspaces, thread-local launch scopes)
   python tools/thread_probe.py [calls per thread] --shared-model   ONE extractor object driven by all threads at once -- two of them
                                                                    on streams of their own, two on the default stream -- with batches
                                                                    of different shapes, some with utterances below MIN_FRAMES (the
                                                                    per-utterance second pass): the reference's layers are stateless after
                                                                    build, so everything a call sets is per thread here
                                                                    (workspace arenas keyed by stream AND thread, pinned routing flag,
                                                                    deferred tail, last_lens)"""
import os, sys, threading, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
import kaldi_tflite_amd as ktf

shared = "--shared-model" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
calls = int(argv[0]) if argv else 40
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=9)
jobs = [("f16mx", 6, 160000), ("bf16x3", 3, 48000), ("f32", 2, 80000), ("f16mx", 1, 160000)]
if shared:                                   # one f16mx model for everybody; MIN_FRAMES stays at its default: ragged 10 s batches have short utterances
    jobs = [("f16mx", 6, 160000), ("f16mx", 3, 48000), ("f16mx", 9, 120000), ("f16mx", 1, 160000)]
wavs = [[torch.as_tensor(synth.make_wav(B, N, seed=100 * j + c, ragged=True), device="cuda") for c in range(4)] for j, (_, B, N) in enumerate(jobs)]
want = []
one = synth.build_extractor(ktf, cfg, w, gemm="f16mx") if shared else None
for j, (mode, B, N) in enumerate(jobs):
    m = one if shared else synth.build_extractor(ktf, cfg, w, gemm=mode)
    want.append([m(x).float().cpu().numpy() for x in wavs[j]])
torch.cuda.synchronize()
errors = []


def worker(j):
    mode, B, N = jobs[j]
    try:
        m = one if shared else synth.build_extractor(ktf, cfg, w, gemm=mode)
        st = torch.cuda.Stream() if (not shared or j < 2) else torch.cuda.current_stream()      # (shared: threads 2 and 3 both sit on the default stream)
        with torch.cuda.stream(st):
            for c in range(calls):
                got = m(wavs[j][c % 4]).float().cpu().numpy()
                if not np.array_equal(got, want[j][c % 4]):
                    errors.append(f"thread {j} ({mode}, B {B}) call {c}: differs from the single-threaded result by {np.abs(got - want[j][c % 4]).max():.3e}")
                    return
    except Exception as e:
        errors.append(f"thread {j} ({mode}): {type(e).__name__}: {e}")


ts = [threading.Thread(target=worker, args=(j,)) for j in range(len(jobs))]
[t.start() for t in ts]
[t.join() for t in ts]
print("\n".join(errors) if errors else f"{len(jobs)} threads x {calls} calls{' on ONE shared model' if shared else ''}: every x-vector equals the single-threaded result")
sys.exit(1 if errors else 0)
