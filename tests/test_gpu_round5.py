"""GPU tests added in round 5 (run with `-m gpu` on an MI355X): one model object under concurrent host threads, the library's build
id and clock probe."""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
from oracle import ktf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_model_object_under_concurrent_host_threads():
    """tools/thread_probe.py --shared-model: FOUR host threads drive ONE XvectorExtractor at once -- two on streams of their own, two on
    the default stream; batches of different shapes, ragged ones with utterances below MIN_FRAMES (the per-utterance second pass and
    its pinned flag) -- and every x-vector equals the single-threaded result bit for bit. The reference's layers are stateless after
    build (kaldi_tflite/lib/layers/tdnn/tdnn.py:251-280, normalization/cmvn.py:186-250): nothing a call sets may live on
    the model object."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "thread_probe.py"), "25", "--shared-model"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "on ONE shared model: every x-vector equals the single-threaded result" in out.stdout, \
        out.stdout[-3000:] + out.stderr[-2000:]


def test_last_lens_and_short_count_are_those_of_the_calling_thread():
    import threading
    cfg, w = synth.extractor_cfg(), synth.make_weights(seed=3)
    m = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
    a = torch.as_tensor(synth.make_wav(2, 160000, seed=1), device="cuda")
    b = torch.as_tensor(synth.make_wav(5, 96000, seed=2), device="cuda")
    m(a)
    seen = {}

    def other():
        assert m.last_lens is None and m.last_short_count == 0          # this thread has not called yet
        m(b)
        seen["shape"] = tuple(m.last_lens.shape)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["shape"] == (5,) and tuple(m.last_lens.shape) == (2,)


def test_build_id_and_clock_probe():
    bid = ops.build_id()
    assert len(bid) == 16 and int(bid, 16) >= 0
    out = torch.zeros(4, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    ops.clock_probe(out, 20000, st)                      # 20 ms on an otherwise idle chip
    torch.cuda.synchronize()
    clk, ticks, lo, hi = [int(v) for v in out.cpu()]
    assert 1_900_000 <= ticks <= 2_600_000               # 100 MHz ticks of 20 ms (+ the loop's last sleep)
    mhz = 100.0 * clk / ticks
    assert 400.0 <= mhz <= 2500.0 and 0 < lo <= hi <= 2_600_000, (mhz, lo, hi)


# ----------------------------------------------------------------------------- pooled layer on flat row tiles (short utterances)
def _pooled_model(U, D, ctx, gemm="bf16x3"):
    cfg = {"type": "sequential", "layers": [
        {"name": "input", "type": "input", "shape": [None, None, D]},
        {"name": "t0", "type": ["affine", "relu", "batchnorm"], "cfg": {"units": D, "context": [0]}},
        {"name": "t", "type": ["affine", "relu", "batchnorm"], "cfg": {"units": U, "context": ctx}},
        {"name": "stats", "type": "stats_pooling", "cfg": {"left_context": 0, "right_context": 5, "include_std": True, "reduce_time_axis": True}}]}
    return ktf.models.SequentialFromConfig(cfg, None, "m", gemm=gemm)


@pytest.mark.parametrize("deterministic", [True, False])
@pytest.mark.parametrize("case", [(1500, 512, [0], 200, 148), (300, 256, [-1, 0, 1], 37, 100), (257, 160, [-2, 0, 2], 700, 23)])
def test_pooled_layer_on_flat_row_tiles_matches_the_per_utterance_tiles_and_the_oracle(case, deterministic):
    """ktf_tdnn_split_flat_stats (csrc/tdnn_split.hip, flat_stats_epilogue): [affine, relu, batchnorm] -> reducing StatsPooling with the
    M-tiles over the batch's valid rows laid end to end. Ragged lengths (empty utterances, single rows, tens of utterances inside one
    128-row block, utterances across tile boundaries) against the per-utterance tiles (same MFMA operands; the fp32 partial sums
    relative to each block's pivot row are cut differently) and against the fp64 oracle; a dead ReLU unit and a constant one give
    std = sqrt(eps) exactly; run-to-run reproducible in the slot form."""
    U, D, ctx, B, T = case
    rng = np.random.default_rng(U + B)
    m = _pooled_model(U, D, ctx)
    W0 = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    W = (rng.standard_normal((U, len(ctx) * D)) / np.sqrt(len(ctx) * D)).astype(np.float32)
    b = (rng.standard_normal(U) * 0.1).astype(np.float32)
    W[5], b[5] = 0.0, -1.0
    W[6], b[6] = 0.0, 0.7
    bn0 = (np.float32(1.0), rng.uniform(0.2, 1.0, D).astype(np.float32), rng.uniform(0.5, 2.0, D).astype(np.float32))
    bn = (np.float32(1.0), rng.uniform(0.2, 1.0, U).astype(np.float32), rng.uniform(0.5, 2.0, U).astype(np.float32))
    m.get_layer("t0.affine").set_weights([W0, np.zeros(D, np.float32)])
    m.get_layer("t0.batchnorm").set_weights(list(bn0))
    m.get_layer("t.affine").set_weights([W, b])
    m.get_layer("t.batchnorm").set_weights(list(bn))
    m.min_tiles, m.min_frames = {}, {}
    m.deterministic = deterministic
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    lens = rng.integers(1, T + 1, B).astype(np.int32)
    lens[[0, B // 2]] = T
    lens[[1, B - 1]] = 0
    lens[2] = 1
    xd, ld = torch.as_tensor(x, device="cuda"), torch.as_tensor(lens, device="cuda")
    seen, orig = [], ops.tdnn_split_flat_stats

    def spy(*a, **k):
        r = orig(*a, **k)
        seen.append(ops.last_kernel())
        return r

    ops.tdnn_split_flat_stats = spy
    try:
        flat = m.run_ragged(xd, ld).float().cpu().numpy()
        flat2 = m.run_ragged(xd, ld).float().cpu().numpy()
    finally:
        ops.tdnn_split_flat_stats = orig
    assert seen == ["tdnn_x3s_kernel<flat, pooled>"] * 2, seen
    m.flat_pooling = False
    tiles = m.run_ragged(xd, ld).float().cpu().numpy()
    ok = lens > 0
    assert np.isnan(flat[~ok][:, 0, :U]).all() and np.isnan(tiles[~ok][:, 0, :U]).all()      # no frame: the mean is 0 / 0, as the reference's pooling gives
    if deterministic:
        assert np.array_equal(flat[ok], flat2[ok])
    scale = np.abs(tiles[ok]).max()
    assert np.abs(flat[ok] - tiles[ok]).max() <= 1e-5 * max(1.0, scale), np.abs(flat[ok] - tiles[ok]).max()     # (fp32 partial sums, cut differently)
    assert np.abs(flat[ok][:, 0, U + 5] - 1e-5).max() < 1e-7 and np.abs(flat[ok][:, 0, U + 6] - 1e-5).max() < 1e-7
    layers = [{"kind": "tdnn", "W": W0, "b": np.zeros(D, np.float32), "context": [0]}, {"kind": "relu"},
              {"kind": "bn", "rms": bn0[0], "mean": bn0[1], "var": bn0[2]},
              {"kind": "tdnn", "W": W, "b": b, "context": ctx}, {"kind": "relu"}, {"kind": "bn", "rms": bn[0], "mean": bn[1], "var": bn[2]},
              {"kind": "stats", "left_context": 0, "right_context": 5, "include_std": True, "reduce_time_axis": True}]
    for i in list(np.flatnonzero(ok)[:6]) + [int(np.flatnonzero(ok)[-1])]:
        want = O.sequential_forward(layers, x[i:i + 1, : lens[i]], dtype=np.float64)
        assert np.abs(flat[i] - want[0]).max() < 2e-4 * max(1.0, np.abs(want).max()), (i, lens[i])


def test_f16mx_layer_behind_a_valid_padded_layer_that_kept_no_row():
    """Found by tools/fuzz_models.py (round 5, seed 606): a VALID-padded layer whose context is longer than what the layers in front of it
    left (12 frames -> 4 after a subsampling layer -> none) hands EMPTY planes to the next f16mx layer; ktf_tdnn_mx took their null
    pointers for a missing argument. Like ktf_tdnn it now returns without a launch: the runner's result has no row, as the oracle's."""
    D = 40
    spec = [(16, [2], "SAME", 1, "affine", None), (256, [4], "SAME", 3, ["affine", "relu"], None), (300, [-3, -1, 1], "VALID", 2, "affine", "relu"),
            (130, [-4], "SAME", 2, "affine", "relu")]
    lcfg = [{"name": "input", "type": "input", "shape": [None, None, D]}]
    for i, (U, ctx, pad, sub, kinds, act) in enumerate(spec):
        c = {"units": U, "context": ctx, "padding": pad, "subsampling_factor": sub}
        if act:
            c["activation"] = act
        lcfg.append({"name": f"t{i}", "type": kinds, "cfg": c})
    mdl = ktf.models.SequentialFromConfig({"type": "sequential", "layers": lcfg}, None, "m", gemm="f16mx")
    mdl.min_tiles, mdl.min_frames = {}, {}
    x = torch.randn((7, 12, D), device="cuda")
    lens = torch.tensor([7, 4, 10, 5, 12, 5, 12], dtype=torch.int32, device="cuda")
    y = mdl.run_ragged(x, lens)
    assert tuple(y.shape) == (7, 0, 130)


def test_split_bf16_of_a_ragged_batch_converts_the_valid_rows_only():
    """ktf_split_bf16_rows: the same hi / lo planes as ktf_split_bf16 on the rows t < lens[b], nothing written beyond them (the f16mx
    model's second pass over a few short utterances used to convert all B x T rows)."""
    rng = np.random.default_rng(5)
    B, T, D = 9, 77, 30
    x = torch.as_tensor(rng.standard_normal((B, T, D)).astype(np.float32) * 100, device="cuda")
    lens = torch.as_tensor(np.array([77, 0, 1, 32, 33, 64, 5, 77, 0], np.int32), device="cuda")
    full = torch.zeros((2, B, T, 32), dtype=torch.bfloat16, device="cuda")
    ops.split_bf16(x, D, full)
    got = torch.full((2, B, T, 32), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.split_bf16(x, D, got, lens)
    for b in range(B):
        n = int(lens[b])
        assert torch.equal(got[:, b, :n], full[:, b, :n]) and bool((got[:, b, n:] == 7.0).all())
    assert float(((full[0].float() + full[1].float())[:, :, :D] - x).abs().max()) <= 2e-4 * 100


@pytest.mark.parametrize("gemm", ["bf16x3", "f16mx", "bf16", "f32"])
def test_dense_call_through_a_valid_padded_layer_that_keeps_no_row(gemm):
    """Found by tools/fuzz_models.py (round 5, seed 99): Sequential.__call__ on a dense batch whose second layer (VALID padding, context
    longer than the four rows the subsampling layer in front left) keeps no row; the layers behind it receive EMPTY tensors, and
    ktf_convert_pad / ktf_split_bf16 / ktf_mx_planes took their null pointers for missing arguments. They return without a launch now, as
    ktf_tdnn does; the result has no row, like the reference's."""
    D = 40
    spec = [(130, [3], "SAME", 3, ["affine", "relu"], None), (16, [0, 4], "VALID", 3, "affine", "tanh"), (300, [-2, 0, 4], "SAME", 1, ["affine", "relu"], None),
            (96, [0, 3], "SAME", 1, ["affine", "relu", "batchnorm"], None)]
    lcfg = [{"name": "input", "type": "input", "shape": [None, None, D]}]
    for i, (U, ctx, pad, sub, kinds, act) in enumerate(spec):
        c = {"units": U, "context": ctx, "padding": pad, "subsampling_factor": sub}
        if act:
            c["activation"] = act
        lcfg.append({"name": f"t{i}", "type": kinds, "cfg": c})
    mdl = ktf.models.SequentialFromConfig({"type": "sequential", "layers": lcfg}, None, "m", gemm=gemm)
    mdl.min_tiles, mdl.min_frames = {}, {}
    x = torch.randn((7, 12, D), device="cuda")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = mdl(x)
        z = mdl.run_ragged(x, torch.full((7,), 12, dtype=torch.int32, device="cuda"))
    assert tuple(y.shape) == (7, 0, 96) and tuple(z.shape) == (7, 0, 96)


@pytest.mark.parametrize("gemm,B,T,spec,pooled", [
    ("bf16x3", 40, 150, [(300, [-3, 2], "SAME", 1, ["affine", "relu", "batchnorm"], None), (512, [3], "VALID", 1, ["affine", "relu"], None),
                         (130, [-2, 1], "SAME", 1, "affine", "relu"), (300, [2, 4], "SAME", 3, "affine", "tanh")], False),
    ("f16mx", 40, 700, [(16, [0, 4], "VALID", 1, "affine", "relu"), (512, [4], "SAME", 1, ["affine", "relu"], None), (300, [-2], "VALID", 1, "affine", None),
                        (256, [1], "SAME", 1, ["affine", "relu"], None)], True)])
def test_flat_row_tiles_behind_a_valid_padded_layer_of_a_dense_batch(gemm, B, T, spec, pooled):
    """Found by tools/fuzz_models.py (round 5, seeds 202 / 203): a DENSE batch (no lengths) whose frame count a VALID-padded layer changed kept
    the flat-row bookkeeping of the old frame count: the flat layers behind it read and wrote the wrong rows (deviation 2e-2 ... 2). The
    bookkeeping is remade when the frame count changes; flat tiles on and off agree bit for bit on frame-level outputs."""
    D = 40
    lcfg = [{"name": "input", "type": "input", "shape": [None, None, D]}]
    for i, (U, ctx, pad, sub, kinds, act) in enumerate(spec):
        c = {"units": U, "context": ctx, "padding": pad, "subsampling_factor": sub}
        if act:
            c["activation"] = act
        lcfg.append({"name": f"t{i}", "type": kinds, "cfg": c})
    if pooled:
        lcfg.append({"name": "stats", "type": "stats", "cfg": {"left_context": 0, "right_context": 10000, "reduce_time_axis": True, "include_std": True}})
    x = torch.randn((B, T, D), generator=torch.Generator().manual_seed(3)).cuda()
    out = {}
    import warnings
    for flat in (True, False):
        mdl = ktf.models.SequentialFromConfig({"type": "sequential", "layers": lcfg}, None, "m", gemm=gemm)
        mdl.min_tiles, mdl.min_frames = {}, {}
        mdl.flat_rows = mdl.flat_rows_long = mdl.mx_flat_rows = flat
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[flat] = mdl.run_ragged(x, None).float().cpu().numpy()
    assert out[True].shape == out[False].shape and np.isfinite(out[False]).all()
    if pooled:
        assert np.abs(out[True] - out[False]).max() <= 1e-5 * max(1.0, np.abs(out[False]).max())
    else:
        assert np.array_equal(out[True], out[False])


# ----------------------------------------------------------------------------- ktf_xvec_post_f32: four embeddings per workgroup beyond 256
def test_xvec_post_same_bits_for_any_batch_split_and_close_to_fp64():
    """Batches beyond one round of workgroups put four embeddings behind every element of the LDA matrix; an embedding's x-vector has the
    same bits whatever batch it arrives in (1, 256 and 1027 at a time), and sits at fp32 rounding distance from an fp64 restatement of
    xvector_extractor.py:174-184 (mean subtraction, LDA + offset, length normalisation)."""
    g = torch.Generator(device="cuda").manual_seed(21)
    for B, din, dout in ((1027, 512, 150), (700, 200, 200), (515, 512, 300)):
        x = torch.randn((B, din), generator=g, device="cuda")
        mean = torch.randn((din,), generator=g, device="cuda") * 0.1
        A = torch.randn((din, dout), generator=g, device="cuda") / din ** 0.5
        off = torch.randn((dout,), generator=g, device="cuda") * 0.01
        whole = ops.xvec_post(x, mean, A, off)
        parts = torch.cat([ops.xvec_post(x[i:i + 256].contiguous(), mean, A, off) for i in range(0, B, 256)])
        assert torch.equal(whole, parts), (B, din, dout)
        ones = torch.cat([ops.xvec_post(x[i:i + 1].contiguous(), mean, A, off) for i in (0, 1, 2, 3, 4, 5, B - 1)])
        assert torch.equal(whole[[0, 1, 2, 3, 4, 5, B - 1]], ones), (B, din, dout)
        x64, A64 = x.double().cpu().numpy(), A.double().cpu().numpy()
        y = (x64 - mean.double().cpu().numpy()) @ A64 + off.double().cpu().numpy()
        y = y / (np.sqrt((y * y).sum(1, keepdims=True)) / np.sqrt(dout))
        assert np.abs(whole.cpu().numpy() - y).max() <= 5e-6, (B, din, dout)


def test_fused_pooling_finalize_keeps_nan_deviation():
    """ktf_stats_finalize / _slots / _flat: mean AND standard deviation of an utterance without a frame are NaN (0 / 0 through tf.nn.relu,
    stats_pooling.py:231-240), and a NaN sum of squares stays NaN; a negative variance from rounding is clamped to zero."""
    D, eps = 8, 1e-5
    sums = torch.zeros((4, 2, D), dtype=torch.float64, device="cuda")
    sums[0, 0], sums[0, 1] = 6.0, 12.5             # three rows: mean 2, E[x^2] - mean^2 = 1/6
    sums[2, 0], sums[2, 1] = 4.0, float("nan")    # NaN activations
    sums[3, 0], sums[3, 1] = 4.0, 7.999999        # two rows of 2.0, variance slightly negative
    lens = torch.tensor([3, 0, 2, 2], dtype=torch.int32, device="cuda")
    out = torch.full((4, 2 * D), 7.0, dtype=torch.float32, device="cuda")
    ops.stats_finalize(sums, lens, 5, D, True, eps, out)
    o = out.cpu().numpy()
    assert np.allclose(o[0, :D], 2.0) and np.allclose(o[0, D:], np.sqrt(12.5 / 3 - 4.0 + eps), rtol=1e-6)
    assert np.isnan(o[1]).all()
    assert np.allclose(o[2, :D], 2.0) and np.isnan(o[2, D:]).all()
    assert np.allclose(o[3, :D], 2.0) and np.allclose(o[3, D:], np.sqrt(eps), rtol=1e-6)
    # the flat form: utterance 1 owns no row of the flat row space
    starts = torch.tensor([0, 3, 3, 5, 7], dtype=torch.int32, device="cuda")
    slots = ops.flat_stats_slots(5)
    fs = torch.zeros((4, slots, 2, D), dtype=torch.float64, device="cuda")
    fs[:, 0] = sums
    out2 = torch.full((4, 2 * D), 7.0, dtype=torch.float32, device="cuda")
    ops.stats_finalize_flat(fs, starts, 5, D, True, eps, out2, slots)
    o2 = out2.cpu().numpy()
    assert np.array_equal(np.isnan(o2), np.isnan(o)) and np.allclose(o2[~np.isnan(o2)], o[~np.isnan(o)])
