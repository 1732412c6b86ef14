#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on one MI355X (development tool; the judged numbers come from bench.py).
usage: python tools/microbench.py [frontend|vadcmvn|stats|gemm|all] [--batch B] [--iters N]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import kaldi_tflite_amd as ktf  # noqa: E402
from kaldi_tflite_amd import _lib as L, ops  # noqa: E402
import synth  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--gemm", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321)
    mdl = synth.build_extractor(ktf, cfg, w, gemm=a.gemm)
    B, N = a.batch, 160000
    g = torch.Generator(device=dev).manual_seed(1)
    wav = torch.clamp(torch.round(1000.0 * torch.randn((B, N), generator=g, device=dev)), -32767, 32767)
    T = mdl.framing.numFrames(N)
    mdl(wav)      # builds workspaces
    D = mdl.mfcc.numMfccs
    fdt = L.act_torch_dtype(mdl.xvec.batch_gemm(B, T))
    W = mdl._ws
    ws = {"mfcc": W.get("mfcc", (B, T, D), torch.float32, dev), "feats": W.get("feats", (B, T, 32), fdt, dev),
          "lens": W.get("lens", (B,), torch.int32, dev), "idx": W.get("idx", (B, T), torch.int32, dev),
          "work": W.get("cmvn_work", (B * T * 2 * D + 2 * D,), torch.float32, dev)}
    if a.what in ("frontend", "all"):
        fr, mf = mdl.framing, mdl.mfcc
        c = L.FrontendCfg.from_buffer_copy(mf._cfg)
        c.frame_size, c.frame_shift = fr.frameWidth, fr.frameShift
        ms = timeit(lambda: ops.frontend(wav, L.IN_WAV, c, mf.tables(dev), L.OUT_MFCC, N, B, T, out=ws["mfcc"]), a.iters)
        print(f"frontend: {ms:.3f} ms  {B*T/ms/1e3:.1f} Mframes/s  {B*T*760/ms/1e6:.0f} GB/s algorithmic")
    if a.what in ("vadcmvn", "all"):
        ms = timeit(lambda: ops.vad_cmvn(ws["mfcc"], mdl.vad.cfg(), mdl.cmvn.cfg(), ws["feats"], ws["lens"], ws["idx"], ws["work"]), a.iters)
        print(f"vad_cmvn: {ms:.3f} ms")
    if a.what in ("gemm", "all"):
        feats = ws["feats"][:, :, :30]
        ms = timeit(lambda: mdl.xvec.run_ragged(feats, ws["lens"]), a.iters)
        print(f"xvec stack (5 GEMM + stats + tdnn6): {ms:.3f} ms")
    if a.what in ("full", "all"):
        ms = timeit(lambda: mdl(wav), a.iters)
        print(f"full step: {ms:.3f} ms  {B/ms*1e3:.0f} utt/s")


if __name__ == "__main__":
    main()
