"""Which kernel each (GEMM mode, layer shape) runs on (run with `-m gpu`): the dispatcher's map is part of the library's
contract -- bench.py's roofline names these kernels, and a kernel generation nothing dispatches to has no place in the library
(tests/test_host_cpu.py::test_library_kernel_families checks the shipped set). ktf_tdnn_last_kernel() reports the family of the
calling thread's last launch."""

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
from kaldi_tflite_amd import _lib as L

pytestmark = pytest.mark.gpu

FRAME_LAYERS = ["5x30->512", "3x512->512", "3x512->512", "1x512->512", "1x512->1500"]


@pytest.fixture(autouse=True, scope="module")
def _reduced_modes_reach_their_kernels():
    S = ktf.models.Sequential           # defaults of models built in this module (instances copy them; no call-time global)
    old = (S.MIN_TILES, S.MIN_FRAMES)
    S.MIN_TILES, S.MIN_FRAMES = {}, {}
    yield
    S.MIN_TILES, S.MIN_FRAMES = old


def trace(mdl, wav):
    """[(layer shape, kernel family)] of every TDNN launch of one extraction."""
    seen = []
    names = ("tdnn", "tdnn_stats", "tdnn_split", "tdnn_split_stats", "tdnn_split_flat", "tdnn_split_flat_stats", "tdnn_mx", "tdnn_mx_flat", "tdnn_mx_stats",
             "tdnn_mx_flat_stats")
    orig = {n: getattr(ops, n) for n in names}

    def wrap(fn):
        def f(x, lens, desc, *a, **k):
            r = fn(x, lens, desc, *a, **k)
            seen.append((f"{int(desc.nctx)}x{int(desc.din)}->{int(desc.units)}", ops.last_kernel()))
            return r
        return f
    for n in names:
        setattr(ops, n, wrap(orig[n]))
    try:
        mdl(torch.as_tensor(wav, device="cuda"))
    finally:
        for n in names:
            setattr(ops, n, orig[n])
    return seen


EXPECT = {   # mode -> kernel family of tdnn1 .. tdnn5 for a batch of full-length utterances
    "f32": ["tdnn_f32t_kernel"] * 5,
    "bf16": ["tdnn_bf16h_kernel", "tdnn_bf16r16_kernel", "tdnn_bf16r16_kernel", "tdnn_bf16h_kernel", "tdnn_bf16h_kernel"],
    "bf16x3": ["tdnn_x3s_kernel<flat>"] * 4 + ["tdnn_x3s_kernel<flat, pooled>"],       # flat row tiles (Sequential.flat_rows_long), as for f16mx
    "f16mx": ["tdnn_mx_kernel<flat>"] * 5,        # 998-frame utterances fill 3.9 of their four 256-row tiles: flat row tiles (Sequential.mx_flat_rows)
}


@pytest.mark.parametrize("gemm", sorted(EXPECT))
def test_batch_dispatch(gemm):
    w = synth.make_weights(seed=4321)
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm=gemm)
    got = trace(mdl, synth.make_wav(32, 160000, seed=3))
    frame = [(s, k) for s, k in got if s in FRAME_LAYERS]
    assert [s for s, _ in frame] == FRAME_LAYERS
    assert [k for _, k in frame] == EXPECT[gemm], frame
    # the affine after the pooling is part of the fused tail kernel (ktf_xvec_tail_f32): no GEMM launch of its own ...
    assert not [k for s, k in got if s == "1x3000->512"], got
    # ... unless the fusion is switched off: exact fp32 in every mode
    mdl.fuse_tail = False
    tail = [k for s, k in trace(mdl, synth.make_wav(32, 160000, seed=3)) if s == "1x3000->512"]
    assert len(tail) == 1 and tail[0].startswith("tdnn_f32"), tail


def test_f16mx_kernel_by_batch_size():
    """An f16mx model as shipped: 6 x 10 s leave the chip three quarters empty on 256-row tiles and run on the loader-wave kernel's
    smaller flat tiles, 32 x 10 s fill it with 256-row tiles (Sequential._mx_use_loader; tools/mid_batch.py)."""
    S = ktf.models.Sequential
    old = (S.MIN_TILES, S.MIN_FRAMES)
    S.MIN_TILES, S.MIN_FRAMES = {"f16mx": 20}, {"f16mx": 400}
    try:
        mdl = synth.build_extractor(ktf, synth.extractor_cfg(), synth.make_weights(seed=4321), gemm="f16mx")
        wav = synth.make_wav(32, 160000, seed=3)
        for B, family in ((6, "tdnn_mxl_kernel"), (32, "tdnn_mx_kernel<flat>")):
            got = [k for _, k in trace(mdl, wav[:B])]
            assert got[:5] == [family] * 5, (B, got)
    finally:
        S.MIN_TILES, S.MIN_FRAMES = old


def test_f16mx_loader_kernel_dispatch():
    """`Sequential.mx_loader`: every frame-level layer of an f16mx model on the loader-wave kernel (csrc/tdnn_mxl.hip)."""
    w = synth.make_weights(seed=4321)
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm="f16mx")
    mdl.xvec.mx_loader = True
    got = [k for s, k in trace(mdl, synth.make_wav(32, 160000, seed=3)) if s in FRAME_LAYERS]
    assert got == ["tdnn_mxl_kernel"] * 5, got


def test_single_utterance_dispatch():
    """One 10 s utterance: the 256-row tiles cannot fill the chip (Sequential.min_tiles). "f32" runs the exact fp32 small-tile kernels;
    the reduced modes run the bf16-pair small tiles (KTF_GEMM_BF16X4) behind an fp32 first layer that writes pairs, or -- with
    `small_tile_pairs` off -- the fp32 kernels throughout."""
    S = ktf.models.Sequential
    old = S.MIN_TILES
    S.MIN_TILES = {"bf16": 6, "bf16x3": 32, "f16mx": 32}
    try:
        w = synth.make_weights(seed=4321)
        wav = synth.make_wav(1, 160000, seed=3)
        for gemm, pairs in (("f32", True), ("bf16x3", True), ("f16mx", True), ("f16mx", False)):
            m1 = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm=gemm)
            m1.fuse_tail = False
            m1.xvec.small_tile_pairs = pairs
            got = trace(m1, wav)
            kernels = [k for _, k in got]
            assert kernels[0] == "tdnn_f32s_kernel<32, 64>"                     # 30 -> 32 input columns: K-step 32
            family = "tdnn_x4s_kernel<64, " if (pairs and gemm != "f32") else "tdnn_f32s_kernel<64, "
            assert all(k.startswith(family) for k in kernels[1:5]), got
            assert kernels[5] == "tdnn_f32_rowvec_kernel"
    finally:
        S.MIN_TILES = old


def test_special_shapes():
    dev = torch.device("cuda")
    x = torch.randn((4, 300, 64), device=dev)
    # fp32 activations handed to a split-bf16 layer: the kernel that splits them in registers
    l = ktf.layers.TDNN(512, context=[-1, 0, 1], name="a", gemm="bf16x3")
    l.build((None, None, 64))
    l(x)
    assert ops.last_kernel() == "tdnn_x3r_kernel"
    # sigmoid on the 16-bit ring: the 32x32x16 kernel
    l = ktf.layers.TDNN(512, context=[-1, 0, 1], activation="sigmoid", name="b", gemm="bf16")
    l.build((None, None, 64))
    l(x)
    assert ops.last_kernel() == "tdnn_bf16r_kernel"
    # a narrow layer (units <= 128)
    l = ktf.layers.TDNN(96, context=[0], name="c", gemm="bf16")
    l.build((None, None, 64))
    l(x)
    assert ops.last_kernel() in ("tdnn_bf16g_kernel", "tdnn_bf16_kernel<64, true, false>", "tdnn_bf16_kernel<64, false, false>")
    # the bitwise reference tiles of the fp32 kernels
    l = ktf.layers.TDNN(512, context=[0], name="d", gemm="f32")
    l.build((None, None, 64))
    l.kernelFlags = L.TDNN_REF_TILES
    l(x)
    assert ops.last_kernel().startswith("tdnn_f32_kernel<")
