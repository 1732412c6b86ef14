#!/usr/bin/env python3
"""Flat row tiles against per-utterance tiles at sizes near the 32-bit offset guards: x-vectors of big batches of long, ragged utterances
(f16mx and bf16x3) with the flat switches on and off -- python tools/flat_big_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import torch
import synth
import kaldi_tflite_amd as ktf
cfg, w = synth.extractor_cfg(), synth.make_weights(seed=4321)
dev = torch.device("cuda")
for gemm, B, sec in (("f16mx", 600, 30.0), ("f16mx", 1024, 30.0), ("bf16x3", 300, 30.0), ("f16mx", 1400, 20.0), ("f16mx", 40, 120.0)):
    g = torch.Generator(device=dev).manual_seed(B)
    n = int(sec * 16000)
    wav = torch.clamp(torch.round(1000.0 * torch.randn((B, n), generator=g, device=dev)), -32767, 32767)
    quiet = torch.rand((B, n // 8000), device=dev, generator=g) < 0.35
    quiet[:, 0] = False
    wav = torch.round(wav * torch.where(quiet, 1e-3, 1.0).repeat_interleave(8000, dim=1))
    out = {}
    for flat in (True, False):
        m = synth.build_extractor(ktf, cfg, w, gemm=gemm)
        m.xvec.mx_flat_rows = m.xvec.flat_rows_long = m.xvec.flat_rows = flat
        out[flat] = m(wav).float().cpu()
        lens = m.last_lens
    ok = torch.isfinite(out[False]).all(1)
    d = (out[True][ok] - out[False][ok]).abs().max().item()
    print(f"{gemm} B={B} {sec:g} s (T = {int(lens.max())} max, {float(lens.float().mean()):.0f} mean voiced): max |flat - tiles| = {d:.2e}; finite rows {int(ok.sum())} == {int(torch.isfinite(out[True]).all(1).sum())}")
    assert d <= 5e-6 and torch.equal(ok, torch.isfinite(out[True]).all(1))
    del wav, m
    torch.cuda.empty_cache()
print("ok")
