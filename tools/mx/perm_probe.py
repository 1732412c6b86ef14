#!/usr/bin/env python3
"""Does the ORDER of the units inside the 256-row f16mx kernel's weight images change its speed? (Round 4 ported the register plane
epilogue to that kernel, which needs the units of every 32-unit chunk permuted in the images, and every layer -- the pooled one too,
whose code was unchanged -- ran 3 % slower.) Timing only: the images are built from row-permuted weights (wrong x-vectors), the kernels
are the product's. python tools/mx/perm_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import synth
import bench
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops, mx

g = torch.Generator(device="cuda").manual_seed(1234)
wav = torch.clamp(torch.round(1000.0 * torch.randn((1024, 160000), generator=g, device="cuda")), -32767, 32767)
orig = mx.weight_images
orders = {"natural": None, "chunk_permuted": mx.loader_unit_order(), "random": np.random.default_rng(5).permutation(256),
          "reversed": np.arange(256)[::-1].copy()}


def build(order):
    def images(Wk):
        if order is not None:
            Up = Wk.shape[0]
            Wk = Wk.reshape((Up // 256, 256) + Wk.shape[1:])[:, order].reshape(Wk.shape)
        return orig(Wk)
    mx.weight_images = images
    m = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f16mx")
    for _ in range(3):
        m(wav)
    mx.weight_images = orig
    return m


models = {k: build(v) for k, v in orders.items()}
for rnd in range(3):
    for k, m in models.items():
        prof = bench._GemmProfiler(ops, torch)
        for _ in range(5):
            m(wav)
        torch.cuda.synchronize()
        st = prof.finish()
        print(f"round {rnd} {k:15s} gemm ms/step {sum(st['per_layer_ms'].values()):.3f} ", {a: round(b, 3) for a, b in st["per_layer_ms"].items()})
        sys.stdout.flush()
