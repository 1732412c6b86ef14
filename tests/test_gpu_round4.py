"""Round-4 GPU tests (run with `-m gpu` on an MI355X): the N > 1 bench path on ONE GPU, and the fused-tail guard.

* `bench.py --gpus 2` is what the driver runs on an 8-GPU node (one rank per GPU over RCCL). No such node is available to the test
  run, so two and four ranks share the one GPU (`KTF_SHARE_GPU=1`) over the gloo backend (`KTF_DIST_BACKEND=gloo`): fresh child processes
  started by bench.py's own self-launch before they touch the GPU. Checked: every rank seen by the collective backend, the gather ran, the
  per-rank step times are reported, and the gathered x-vectors are each rank's own, bit for bit (BASELINE.json config 4: batch-sharded utterances, gather of
  the embeddings).
* an extractor whose LDA keeps more than 256 dimensions takes the three-launch tail (ktf_xvec_tail_f32 serves out_dim <= 256).
"""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
from kaldi_tflite_amd import layers as Ls
from oracle import ktf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("N", [2, 4])
def test_bench_ranks_on_one_gpu_gather_every_shard(tmp_path, N):
    B, sec = 48, 3.0
    dump = str(tmp_path / "xv.npy")
    env = dict(os.environ, KTF_SHARE_GPU="1", KTF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline",
           "--no-parity", "--batch", str(B), "--seconds", str(sec), "--gemm", "f32", "--dump-xvectors", dump]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == N and line["scaling"] == "weak"
    assert line["config"]["ranks_seen_by_collective_backend"] == N and line["config"]["gather"] is True
    pr = line["per_rank_ms_per_step"]
    assert pr["ranks"] == N and 0 < pr["min"] <= pr["max"] and abs(pr["max"] - line["ms_per_step"]) < 1e-6 * pr["max"]
    assert line["config"]["collective_backend"] == "gloo"
    assert line["value"] > 0 and abs(line["value"] - N * B / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    got = np.load(dump)
    assert got.shape == (N * B, 128)
    # the same extraction in this process: rank r's waveforms come from generator seed 1234 + r (bench.py)
    cfg = synth.extractor_cfg(dither=0.0)
    mdl = synth.build_extractor(ktf, cfg, synth.make_weights(seed=4321, narrow=False), gemm="f32")
    for r in range(N):
        g = torch.Generator(device="cuda").manual_seed(1234 + r)
        wav = torch.clamp(torch.round(1000.0 * torch.randn((B, int(sec * 16000)), generator=g, device="cuda")), -32767, 32767)
        want = mdl(wav).cpu().numpy()
        assert np.array_equal(got[r * B:(r + 1) * B], want), f"rank {r}'s shard of the gathered embeddings"


def test_wide_lda_takes_the_three_launch_tail():
    """LDA output wider than the fused tail serves (300 > 256): finalize + GEMM + ktf_xvec_post_f32, same x-vectors as the oracle."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=5)
    rng = np.random.default_rng(17)
    w["lda"] = (rng.standard_normal((300, 513)) / np.sqrt(512)).astype(np.float32)
    wav = synth.make_wav(3, 40000, seed=9, ragged=True)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    for gemm in ("f32", "f16mx"):
        mdl = synth.build_extractor(ktf, cfg, w, gemm=gemm)
        mdl.xvec.min_tiles, mdl.xvec.min_frames = {}, {}
        assert not mdl._tail_fusable()
        got = mdl(torch.as_tensor(wav, device="cuda"))
        assert ops.last_kernel().startswith("tdnn_f32"), ops.last_kernel()          # tdnn6 ran as a GEMM launch of its own
        assert got.shape == (3, 300)
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-4
    narrow = synth.build_extractor(ktf, cfg, synth.make_weights(seed=5), gemm="f32")
    assert narrow._tail_fusable()


# ----------------------------------------------------------------------------- bf16-pair small tiles (KTF_GEMM_BF16X4)
def _pair_case(rng, B, T, D, U, ctx, sub, pad, relu, lens=None):
    from kaldi_tflite_amd import ops
    x = rng.standard_normal((B, T, D)).astype(np.float32) * 3.0
    W = (rng.standard_normal((U, len(ctx) * D)) / np.sqrt(len(ctx) * D)).astype(np.float32)
    b = rng.standard_normal(U).astype(np.float32)
    t = ktf.layers.TDNN(U, context=list(ctx), subsampling_factor=sub, padding=pad, name="p")
    t.build(x.shape)
    t.set_weights([W, b])
    Dp = ops.round_up(D, 32)
    xp = torch.zeros((B, T, Dp), device="cuda")
    xp[:, :, :D] = torch.as_tensor(x, device="cuda")
    return t, x, W, b, xp


@pytest.mark.parametrize("case", [
    (1, 998, 512, 512, [-2, 0, 2], 1, "SAME", True),          # tdnn2 of one utterance: 64 x 32 tiles
    (1, 998, 512, 1500, [0], 1, "SAME", True),                # tdnn5: 64 x 96 tiles
    (3, 200, 512, 512, [0], 1, "SAME", False),                # 64 x 64 tiles
    (2, 141, 40, 33, [-3, 1], 2, "VALID", True),              # din_pad 64; VALID + subsampling; pad units
    (2, 90, 96, 200, [-1, 0, 1], 1, "SAME", False),           # din_pad 96: K-step 32
])
def test_pair_kernel_vs_oracle(case):
    """ktf_tdnn with KTF_GEMM_BF16X4: operands as bf16 pairs (16 mantissa bits), all four partial products, fp32 accumulation:
    against the fp64 oracle ON THE SAME pair-rounded operands the error is fp32 summation noise; against the exact operands it is
    the 2^-17 of the split. Ragged lengths, VALID padding, subsampling; rows beyond an utterance's output stay untouched; a pair
    output decodes to the fp32 output within 2^-17."""
    from kaldi_tflite_amd import ops, _lib as L
    from oracle import ktf_oracle as O
    rng = np.random.default_rng(77)
    B, T, D, U, ctx, sub, pad, relu = case
    t, x, W, b, xp = _pair_case(rng, *case)
    lens = [T] + [max(1, T // 3)] * (B - 1)
    dl = torch.as_tensor(np.array(lens, np.int32), device="cuda")
    Tout = t.outputTimesteps(T)
    sc = rng.uniform(0.5, 2.0, U).astype(np.float32)
    sh = rng.uniform(-1.0, 1.0, U).astype(np.float32)
    bn = (torch.as_tensor(sc, device="cuda"), torch.as_tensor(sh, device="cuda"))
    pairs = ops.pair_encode(xp)
    out = torch.full((B, Tout, U), 7.0, device="cuda")
    out_lens = torch.zeros(B, dtype=torch.int32, device="cuda")
    t.forward(pairs, lens=dl, relu=relu, bn=bn, gemm=L.GEMM_BF16X4, out=out, out_lens=out_lens, pair_in=True)
    assert ops.last_kernel().startswith("tdnn_x4s_kernel")
    got = out.cpu().numpy()
    xr = ops.pair_decode(pairs).cpu().numpy()[:, :, :D].astype(np.float64)
    Wr = ops.pair_decode(ops.pair_encode(torch.as_tensor(W))).numpy().astype(np.float64)
    for bi in range(B):
        want_r = O.tdnn(xr[bi:bi + 1, : lens[bi]], Wr, b, ctx, sub, pad, "relu" if relu else None, dtype=np.float64)[0] * sc + sh
        want_x = O.tdnn(x[bi:bi + 1, : lens[bi]], W, b, ctx, sub, pad, "relu" if relu else None, dtype=np.float64)[0] * sc + sh
        n = want_r.shape[0]
        assert int(out_lens[bi]) == n
        scale = max(1.0, np.abs(want_x).max(initial=0.0))
        assert np.abs(got[bi, :n] - want_r).max(initial=0.0) <= 3e-6 * scale, "kernel != its own arithmetic"
        assert np.abs(got[bi, :n] - want_x).max(initial=0.0) <= 4e-5 * scale
        assert (got[bi, n:] == 7.0).all()
    # pair output of the pair kernel and of the fp32 kernel
    outp = torch.zeros((B, Tout, U), device="cuda")
    t.forward(pairs, lens=dl, relu=relu, bn=bn, gemm=L.GEMM_BF16X4, out=outp, pair_in=True, pair_out=True)
    dec = ops.pair_decode(outp).cpu().numpy()
    out32 = torch.zeros((B, Tout, U), device="cuda")
    t.forward(xp, lens=dl, relu=relu, bn=bn, gemm=L.GEMM_F32, out=out32)
    outp32 = torch.zeros((B, Tout, U), device="cuda")
    t.forward(xp, lens=dl, relu=relu, bn=bn, gemm=L.GEMM_F32, out=outp32, pair_out=True)
    dec32 = ops.pair_decode(outp32).cpu().numpy()
    ref32 = out32.cpu().numpy()
    for bi in range(B):
        n = int(out_lens[bi])
        assert np.abs(dec[bi, :n] - got[bi, :n]).max(initial=0.0) <= 2.0 ** -16 * max(1.0, np.abs(got[bi, :n]).max(initial=0.0))
        assert np.abs(dec32[bi, :n] - ref32[bi, :n]).max(initial=0.0) <= 2.0 ** -16 * max(1.0, np.abs(ref32[bi, :n]).max(initial=0.0))


def test_pair_kernel_random_shapes_against_the_fp32_kernels():
    """Forty random layers (units 1 ... 700, input width 1 ... 300, 1-5 context offsets within +-6, SAME / VALID, subsampling 1-3, ragged
    batches with empty and one-frame utterances, fused ReLU and BatchNorm affine on or off): the pair kernel against the exact fp32
    kernels on the same fp32 inputs -- the 2^-17 operand split is the only difference -- and the same out_lens."""
    from kaldi_tflite_amd import ops, _lib as L
    rng = np.random.default_rng(2024)
    for trial in range(40):
        B = int(rng.integers(1, 5))
        T = int(rng.integers(1, 400))
        D = int(rng.integers(1, 301))
        U = int(rng.integers(1, 701))
        K = int(rng.integers(1, 6))
        ctx = sorted(rng.choice(np.arange(-6, 7), size=K, replace=False).tolist())
        sub = int(rng.integers(1, 4))
        pad = "VALID" if rng.random() < 0.4 else "SAME"
        relu = bool(rng.random() < 0.5)
        t, x, W, b, xp = _pair_case(rng, B, T, D, U, ctx, sub, pad, relu)
        lens = rng.integers(0, T + 1, size=B)
        lens[0] = T
        if B > 1:
            lens[1] = min(T, 1)
        dl = torch.as_tensor(lens.astype(np.int32), device="cuda")
        bn = None
        if rng.random() < 0.5:
            bn = (torch.as_tensor(rng.uniform(0.5, 2.0, U).astype(np.float32), device="cuda"),
                  torch.as_tensor(rng.uniform(-1.0, 1.0, U).astype(np.float32), device="cuda"))
        Tout = t.outputTimesteps(T)
        if Tout == 0:
            continue
        got = torch.full((B, Tout, U), 7.0, device="cuda")
        ref = torch.full((B, Tout, U), 7.0, device="cuda")
        gl = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        rl = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        t.forward(ops.pair_encode(xp), lens=dl, relu=relu, bn=bn, gemm=L.GEMM_BF16X4, out=got, out_lens=gl, pair_in=True)
        t.forward(xp, lens=dl, relu=relu, bn=bn, gemm=L.GEMM_F32, out=ref, out_lens=rl)
        assert torch.equal(gl, rl), (trial, gl, rl)
        g, r = got.cpu().numpy(), ref.cpu().numpy()
        for bi in range(B):
            n = int(rl[bi])
            scale = max(1.0, float(np.abs(r[bi, :n]).max(initial=0.0)))
            assert np.abs(g[bi, :n] - r[bi, :n]).max(initial=0.0) <= 5e-5 * scale, (trial, B, T, D, U, ctx, sub, pad, relu)
            assert (g[bi, n:] == 7.0).all(), (trial, "rows beyond the utterance")


@pytest.mark.parametrize("units,ctx", [(1500, [0]), (512, [-2, 0, 2]), (200, [0])])
def test_pair_kernel_fused_pooling(units, ctx):
    """ktf_tdnn_stats with KTF_GEMM_BF16X4: fp64 column sums per 64-row tile (KTF_TDNN_DET_STATS: ktf_tdnn_stats_slots slots of
    ktf_tdnn_slot_rows rows) or by atomics, finalized to mean | std; against the same layer written out and pooled on the host."""
    from kaldi_tflite_amd import ops, _lib as L
    rng = np.random.default_rng(78)
    B, T, D = 3, 300, 512
    t, x, W, b, xp = _pair_case(rng, B, T, D, units, ctx, 1, "SAME", True)
    lens = [300, 129, 64]
    dl = torch.as_tensor(np.array(lens, np.int32), device="cuda")
    sc = rng.uniform(0.5, 2.0, units).astype(np.float32)
    sh = rng.uniform(-1.0, 1.0, units).astype(np.float32)
    bn = (torch.as_tensor(sc, device="cuda"), torch.as_tensor(sh, device="cuda"))
    pairs = ops.pair_encode(xp)
    rows = torch.zeros((B, T, units), device="cuda")
    t.forward(pairs, lens=dl, relu=True, bn=bn, gemm=L.GEMM_BF16X4, out=rows, pair_in=True)
    y = rows.cpu().numpy().astype(np.float64)
    w, _, bias = t.device_weights(torch.device("cuda"), L.GEMM_BF16X4)
    assert ops.tdnn_slot_rows(L.GEMM_BF16X4) == 64 and ops.tdnn_stats_slots(T, L.GEMM_BF16X4) == 5
    for det in (True, False):
        slots = ops.tdnn_stats_slots(T, L.GEMM_BF16X4) if det else 0
        d = t.desc(L.GEMM_BF16X4, L.PAIR, torch.float32, act="relu", flags=L.TDNN_DET_STATS if det else 0)
        sums = torch.full((B, max(slots, 1), 2, units), 3.0, dtype=torch.float64, device="cuda")
        ops.tdnn_stats(pairs, dl, d, w, None, bias, bn[0], bn[1], sums, zero=not det)
        assert ops.last_kernel().startswith("tdnn_x4s_kernel")
        out = torch.zeros((B, 2 * units), device="cuda")
        ops.stats_finalize(sums, dl, T, units, True, 1e-10, out, slots=slots, slot_rows=64)
        got = out.cpu().numpy()
        for bi in range(B):
            v = y[bi, : lens[bi]]
            want = np.concatenate([v.mean(0), np.sqrt(np.maximum((v * v).mean(0) - v.mean(0) ** 2, 0) + 1e-10)])
            assert np.abs(got[bi] - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (det, bi)


@pytest.mark.parametrize("gemm", ["f16mx", "bf16x3", "bf16"])
def test_small_batches_on_the_pair_route_vs_oracle(gemm):
    """Batches below Sequential.min_tiles of every reduced mode: x-vectors through the pair route against the fp64 oracle (the bar
    of the compliant modes: 1e-4; measured ~1e-5), for one utterance and for a ragged handful, and the route switched off."""
    from oracle import ktf_oracle as O
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321)
    wav = np.concatenate([synth.make_wav(1, 160000, seed=3), synth.make_wav(2, 160000, seed=5, ragged=True)], 0)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    mdl = synth.build_extractor(ktf, cfg, w, gemm=gemm)
    assert mdl.xvec.min_tiles.get(gemm, 0) > 3
    for sl in (slice(0, 1), slice(0, 3)):
        if sl.stop * 4 >= mdl.xvec.min_tiles[gemm]:        # (10 s = four 256-row tiles per utterance: "bf16" fills the chip from 6 tiles on)
            continue
        got = mdl(torch.as_tensor(wav[sl], device="cuda")).cpu().numpy()
        err = np.abs(got - want[sl]).max()
        print(f"pair route, {gemm}, batch {sl.stop}: max-abs dev vs fp64 oracle {err:.2e}")
        assert err <= 3e-5
    mdl.xvec.small_tile_pairs = False
    got = mdl(torch.as_tensor(wav[:1], device="cuda")).cpu().numpy()
    assert np.abs(got - want[:1]).max() <= 2e-5


def test_captured_graph_of_a_routed_batch():
    """XvectorExtractor.compile on a batch the per-utterance routing applies to (f16mx, 256-row kernels): the capture holds BOTH passes --
    the second one's weights and workspaces exist before the capture starts (they used to be uploaded on first use, i.e. inside the
    capture: hipErrorStreamCaptureUnsupported) -- and a replay equals the eager call bit for bit, with and without a short utterance."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321)
    wav = synth.make_wav(6, 160000, seed=11)
    mdl = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
    assert mdl.xvec.batch_gemm(6, 998) == ktf._lib.GEMM_F16MX and mdl.route_short_utterances
    x = torch.as_tensor(wav, device="cuda")
    run = mdl.compile(x)
    y = mdl(x)
    assert mdl.last_short_count == 0
    assert torch.equal(run(x), y)
    quiet = wav.copy()
    quiet[2, 30000:] = 0.0                              # utterance 2: < 400 voiced frames -> the split-bf16 pass
    xq = torch.as_tensor(quiet, device="cuda")
    yq = mdl(xq)
    assert mdl.last_short_count == 1
    assert torch.equal(run(xq), yq)
    assert not torch.equal(yq[2], y[2]) and torch.equal(yq[0], y[0])


# ----------------------------------------------------------------------------- flat row tiles of the split-bf16 plane kernel
@pytest.mark.parametrize("case", [
    (40, 148, 512, 512, [-2, 0, 2], "1.5 s windows"),
    (7, 300, 512, 1500, [0], "3 s, an N-tile with pad units"),
    (700, 5, 512, 256, [-1, 0, 1], "hundreds of utterances inside one tile, empty ones among them"),
])
def test_flat_row_tiles_equal_the_per_utterance_tiles_bit_for_bit(case):
    """ktf_tdnn_split_flat lays the M-tiles over the batch's valid rows end to end; every row still clamps its context offsets against
    its own utterance and goes through the same MFMAs in the same order: planes and fp32 rows equal ktf_tdnn_split's bit for bit on
    every valid row, and rows beyond an utterance's length are not written."""
    from kaldi_tflite_amd import _lib as L
    B, T, D, U, ctx, _ = case
    rng = np.random.default_rng(123)
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    W = (rng.standard_normal((U, len(ctx) * D)) / np.sqrt(len(ctx) * D)).astype(np.float32)
    b = rng.standard_normal(U).astype(np.float32)
    t = ktf.layers.TDNN(U, context=list(ctx), name="f")
    t.build(x.shape)
    t.set_weights([W, b])
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    lens[0] = T
    lens[B // 2] = 0
    dl = torch.as_tensor(lens, device="cuda")
    planes = torch.zeros((2, B, T, D), dtype=torch.bfloat16, device="cuda")
    ops.split_bf16(torch.as_tensor(x, device="cuda"), D, planes)
    dev = torch.device("cuda")
    w, w_lo, bias = t.device_weights(dev, L.GEMM_BF16X3, k_interleaved=len(ctx) > 1, w_tiled=True)
    kflag = (L.TDNN_K_INTERLEAVED if len(ctx) > 1 else 0) | L.TDNN_W_TILED
    sc = torch.as_tensor(rng.uniform(0.5, 2.0, U).astype(np.float32), device="cuda")
    sh = torch.as_tensor(rng.uniform(-1.0, 1.0, U).astype(np.float32), device="cuda")
    starts = ops.row_starts(dl, B, T, torch.zeros(B + 1, dtype=torch.int32, device="cuda"))
    assert starts.tolist() == [0] + np.cumsum(lens).tolist()
    ldy = ops.round_up(U, 32)
    for out_planes in (True, False):
        ydt = torch.bfloat16 if out_planes else torch.float32
        d = t.desc(L.GEMM_BF16X3, torch.bfloat16, ydt, act="relu", flags=kflag)
        shape = (2, B, T, ldy) if out_planes else (B, T, ldy)
        ref = torch.full(shape, 7.0, dtype=ydt, device="cuda")
        got = torch.full(shape, 7.0, dtype=ydt, device="cuda")
        if out_planes:
            ops.tdnn_split(planes, dl, d, w, w_lo, bias, sc, sh, ref[0], ref[1])
            assert ops.last_kernel() == "tdnn_x3s_kernel"
            ops.tdnn_split_flat(planes, starts, d, w, w_lo, bias, sc, sh, got[0], got[1])
        else:
            ops.tdnn_split(planes, dl, d, w, w_lo, bias, sc, sh, ref, None)
            ops.tdnn_split_flat(planes, starts, d, w, w_lo, bias, sc, sh, got, None)
        assert ops.last_kernel() == "tdnn_x3s_kernel<flat>"
        assert torch.equal(got, ref), case[-1]
        g = got.float().cpu().numpy()
        g = g if out_planes else g[None]
        for bi in (0, B // 2, B - 1):
            assert (g[:, bi, lens[bi]:, :U] == 7.0).all()


def test_flat_row_tiles_random_batches():
    """Twenty-five random ragged batches (1 ... 300 utterances of up to 20 ... 400 frames, empty ones and one-frame ones among them,
    1-5 context offsets within +-9, input widths 40 ... 512, 129 ... 700 units): flat row tiles == per-utterance tiles, bit for bit,
    and nothing is written beyond an utterance's length."""
    from kaldi_tflite_amd import _lib as L
    rng = np.random.default_rng(4242)
    dev = torch.device("cuda")
    for trial in range(25):
        B = int(rng.integers(1, 301))
        T = int(rng.integers(20, 401))
        D = int(rng.choice([40, 64, 96, 200, 512]))
        U = int(rng.integers(129, 701))
        K = int(rng.integers(1, 6))
        ctx = sorted(rng.choice(np.arange(-9, 10), size=K, replace=False).tolist())
        relu = bool(rng.random() < 0.5)
        x = rng.standard_normal((B, T, D)).astype(np.float32)
        W = (rng.standard_normal((U, K * D)) / np.sqrt(K * D)).astype(np.float32)
        t = ktf.layers.TDNN(U, context=ctx, name="r")
        t.build(x.shape)
        t.set_weights([W, rng.standard_normal(U).astype(np.float32)])
        lens = rng.integers(0, T + 1, size=B).astype(np.int32)
        lens[rng.integers(0, B)] = T
        lens[rng.integers(0, B)] = min(T, 1)
        dl = torch.as_tensor(lens, device="cuda")
        Dp = ops.round_up(D, 32)
        planes = torch.zeros((2, B, T, Dp), dtype=torch.bfloat16, device="cuda")
        ops.split_bf16(torch.as_tensor(x, device="cuda"), D, planes)
        kint = K > 1
        w, w_lo, bias = t.device_weights(dev, L.GEMM_BF16X3, k_interleaved=kint, w_tiled=True)
        kflag = (L.TDNN_K_INTERLEAVED if kint else 0) | L.TDNN_W_TILED
        starts = ops.row_starts(dl, B, T, torch.zeros(B + 1, dtype=torch.int32, device="cuda"))
        ldy = ops.round_up(U, 32)
        d = t.desc(L.GEMM_BF16X3, torch.bfloat16, torch.bfloat16, act="relu" if relu else None, flags=kflag)
        ref = torch.full((2, B, T, ldy), 7.0, dtype=torch.bfloat16, device="cuda")
        got = torch.full((2, B, T, ldy), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.tdnn_split(planes, dl, d, w, w_lo, bias, None, None, ref[0], ref[1])
        ops.tdnn_split_flat(planes, starts, d, w, w_lo, bias, None, None, got[0], got[1])
        assert torch.equal(got, ref), (trial, B, T, D, U, ctx)
        # ... and with the row table made once per batch (ktf_flat_row_map) instead of derived by every workgroup: the same rows
        fr = ops.flat_rows(dl, B, T, lambda role, shape, dt: torch.zeros(shape, dtype=dt, device="cuda"))
        rs = np.concatenate([[0], np.cumsum(lens)])
        want = np.tile(np.array([[-1, 0, 1, 0]], np.int32), (fr.map.shape[0], 1))
        for bi in range(B):
            r = np.arange(rs[bi], rs[bi + 1])
            want[r] = np.stack([bi * T + (r - rs[bi]), r - rs[bi], np.full(len(r), lens[bi]), np.full(len(r), bi)], 1)
        assert np.array_equal(fr.map.cpu().numpy(), want), (trial, B, T)
        got2 = torch.full((2, B, T, ldy), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.tdnn_split_flat(planes, fr, d, w, w_lo, bias, None, None, got2[0], got2[1])
        assert torch.equal(got2, ref), (trial, B, T, D, U, ctx)


def test_short_windows_run_flat_rows_and_equal_the_tiled_route():
    """A split-bf16 model on 1.5 s windows (what an f16mx model routes its short utterances to): the plane layers run on flat row
    tiles (Sequential.flat_rows), the pooled layer included (round 5: its partial sums are cut along the flat row space, so the
    x-vectors agree with the per-utterance tiles' to the last bits of the fp32 partial sums, not bit for bit; the frame-level layers do:
    test_flat_row_tiles_equal_the_per_utterance_tiles_bit_for_bit above); 10 s utterances keep the tiles with
    `flat_rows_long` off (and take the flat ones, like the f16mx layers, with it on: tests/test_gpu_dispatch.py)."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321)
    wav = synth.make_wav(48, 24000, seed=21, ragged=True)
    got = {}
    for flat in (True, False):
        mdl = synth.build_extractor(ktf, cfg, w, gemm="bf16x3")
        mdl.xvec.min_tiles = {}
        mdl.xvec.flat_rows = flat
        assert mdl.xvec.flat_pooling
        seen = []
        real = (ops.tdnn_split_flat, ops.tdnn_split_flat_stats)

        def spy(fn):
            def run(*a, **k):
                seen.append(fn.__name__)
                return fn(*a, **k)
            return run
        ops.tdnn_split_flat, ops.tdnn_split_flat_stats = spy(real[0]), spy(real[1])
        try:
            got[flat] = mdl(torch.as_tensor(wav, device="cuda"))
        finally:
            ops.tdnn_split_flat, ops.tdnn_split_flat_stats = real
        assert seen == (["tdnn_split_flat"] * 4 + ["tdnn_split_flat_stats"] if flat else [])       # tdnn1-4, tdnn5 + pooling
    assert float((got[True] - got[False]).abs().max()) <= 2e-6
    mdl = synth.build_extractor(ktf, cfg, w, gemm="bf16x3")
    mdl.xvec.min_tiles = {}
    mdl.xvec.flat_rows_long = False                     # (round 5: long batches take the flat tiles too unless this is off)
    calls = []
    real = ops.tdnn_split_flat
    ops.tdnn_split_flat = lambda *a, **k: calls.append(1) or real(*a, **k)
    try:
        mdl(torch.as_tensor(synth.make_wav(16, 160000, seed=22), device="cuda"))
    finally:
        ops.tdnn_split_flat = real
    assert not calls


# ----------------------------------------------------------------------------- fused VAD / CMVN: utterances split over workgroups
def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_fused_vad_cmvn_split_over_workgroups_matches_the_oracle(out_dtype):
    """Batches below 256 utterances spread each utterance over up to eight workgroups (csrc/vad_cmvn.hip: `gridDim.y` splits by
    window-start chunks); batches of 256 and more run one workgroup per utterance. Both against vad.py:156-203 -> gather ->
    cmvn.py:186-250 (oracle), over windows shorter / longer than the utterance, VALID and SAME, norm_vars, utterance lengths
    around the chunk and window sizes, and: a split batch equals the same utterances inside a big batch bit for bit."""
    rng = np.random.default_rng(77)
    D = 30
    vcfg = dict(energy_mean_scale=0.5, energy_threshold=5.5, frames_context=2, proportion_threshold=0.12)
    tol = {torch.float32: 2e-4, torch.bfloat16: 1.5e-1}[out_dtype]
    for (T, window, nv, pad) in [(998, 300, False, "SAME"), (998, 300, True, "SAME"), (700, 300, False, "VALID"), (335, 300, True, "VALID"),
                                 (333, 300, False, "SAME"), (250, 300, False, "SAME"), (64, 300, True, "SAME"), (1500, 64, False, "SAME"),
                                 (401, 201, True, "VALID")]:
        B = 5
        feats = (rng.standard_normal((B, T, D)) * 4 + 6).astype(np.float32)
        feats[1, T // 3:, 0] = -50.0                       # an utterance that loses two thirds of its frames
        feats[2, :, 0] = -50.0                             # ... and one that loses all of them
        ccfg = Ls.CMVN(window=window, norm_vars=nv, padding=pad).cfg()

        def run(f):
            n = f.shape[0]
            out = torch.full((n, T, 32), 9.0, device="cuda", dtype=out_dtype)
            lens = torch.zeros((n,), dtype=torch.int32, device="cuda")
            idx = torch.zeros((n, T), dtype=torch.int32, device="cuda")
            work = torch.zeros((n * T * 2 * D + 64,), device="cuda")
            ktf.ops.vad_cmvn(dev(f), Ls.VAD(**vcfg).cfg(), ccfg, out, lens, idx, work)
            return out.float().cpu().numpy(), lens.cpu().numpy(), idx.cpu().numpy()

        got, lens, idx = run(feats)                        # 5 utterances: eight workgroups each
        big = np.concatenate([feats, np.repeat(feats[:1], 256, axis=0)])
        got_big, lens_big, _ = run(big)                    # 261: one workgroup each
        assert np.array_equal(lens, lens_big[:B])
        keep = O.vad(feats, **vcfg, return_indexes=False)[..., 0] > 0
        for b in range(B):
            sel = np.nonzero(keep[b])[0]
            want = O.cmvn(feats[b:b + 1, sel], norm_vars=nv, window=window, padding=pad, dtype=np.float64)[0] if len(sel) else np.zeros((0, D))
            n = want.shape[0]
            assert lens[b] == n, (T, window, nv, pad, b, lens[b], n)
            assert np.array_equal(idx[b, : len(sel)], sel)
            assert np.array_equal(got[b, :n], got_big[b, :n]), "split workgroups == one workgroup, bit for bit"
            if n:
                assert np.abs(got[b, :n, :D] - want).max() < tol * max(1.0, np.abs(want).max()), (T, window, nv, pad, b)
                assert not got[b, :n, D:].any(), "pad columns are written as zeros"
            assert (got[b, n:] == 9.0).all(), "rows beyond the utterance's output length are not written"


def test_cmvn_valid_padding_of_inputs_no_longer_than_the_window():
    """cmvn.py:238-243 (and its TODO): VALID padding keeps the frames [N/2, T - (N-1)/2) -- none when T < N, one when T == N -- normalised
    with the whole input's statistics. (The kernels used to return all T rows in that case.)"""
    rng = np.random.default_rng(3)
    for N in (300, 301):
        for T in (1, N // 2, N - 1, N, N + 1):
            x = rng.standard_normal((2, T, 30)).astype(np.float32) * 3 + 1
            for nv in (False, True):
                got = Ls.CMVN(window=N, norm_vars=nv, padding="VALID")(dev(x)).cpu().numpy()
                want = O.cmvn(x, norm_vars=nv, window=N, padding="VALID", dtype=np.float64)
                assert got.shape == want.shape == (2, max(T - N + 1, 0), 30), (N, T, got.shape, want.shape)
                if want.size:
                    assert np.abs(got - want).max() < 1e-4


# ----------------------------------------------------------------------------- found by differential fuzzing (tools/fuzz_layers.py)
def test_single_frame_inputs_with_a_shifted_context_stay_inside_their_utterance():
    """TDNN(context=[c != 0]) on a batch of one-frame inputs: SAME padding clamps every offset to the utterance's only frame
    (tdnn.py:246-247). The layer's 'one B-row GEMM' short cut for T == 1 used to apply the offset across the batch."""
    rng = np.random.default_rng(0)
    for gemm, tol in (("f32", 1e-5), ("bf16x3", 1e-4), ("f16mx", 2e-3)):       # (16 units: an f16mx layer this narrow runs in fp32)
        for ctx, U in (([6], 16), ([-3], 16), ([3], 300), ([0, 6], 16)):
            x = rng.standard_normal((3, 1, 24)).astype(np.float32)
            W = (rng.standard_normal((U, len(ctx) * 24)) / np.sqrt(len(ctx) * 24)).astype(np.float32)
            b = rng.standard_normal(U).astype(np.float32)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t = Ls.TDNN(U, context=list(ctx), gemm=gemm)
                t.build(x.shape)
                t.set_weights([W, b])
                got = t(dev(x)).cpu().numpy()
            want = O.tdnn(x, W, b, ctx, 1, "SAME", None, dtype=np.float64)
            assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), (gemm, ctx, U)


def test_layers_without_an_output_row_and_degenerate_statistics_follow_the_reference():
    rng = np.random.default_rng(1)
    # VALID padding of an input shorter than the context: an empty output, in every mode (the f16mx layer used to raise)
    x = rng.standard_normal((2, 5, 64)).astype(np.float32)
    for gemm in ("f32", "bf16x3", "f16mx"):
        t = Ls.TDNN(256, context=[-5, -3, 3], padding="VALID", activation="relu", gemm=gemm)
        t.build(x.shape)
        assert tuple(t(dev(x)).shape) == (2, 0, 256)
    # one frame with norm_vars: the reference's variance is exactly 0 and its output 0 / 0 (cmvn.py:222-246)
    one = rng.standard_normal((2, 1, 30)).astype(np.float32)
    assert np.isnan(Ls.CMVN(window=100, norm_vars=True)(dev(one)).cpu().numpy()).all()
    assert np.isnan(O.cmvn(one, norm_vars=True, window=100)).all()
    # a pooling window without a sampled frame (input_period 3 on one frame): mean 0 / 0, and tf.nn.relu keeps the NaN of the variance
    sp = dict(left_context=-7, right_context=10, input_period=3, output_period=6, include_std=True, padding="SAME")
    got = Ls.StatsPooling(**sp)(dev(one)).cpu().numpy()
    want = O.stats_pooling(one, **sp, dtype=np.float64)
    assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).any()


def test_differential_fuzzing_of_the_layer_api_finds_nothing():
    """tools/fuzz_layers.py: random StatsPooling / CMVN / TDNN / VAD / Framing / MFCC configurations (degenerate lengths included)
    against the oracle; a fixed seed here, more rounds by hand."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_layers.py"), "120", "11"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "120 rounds, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def _stack(spec, D, gemm, rng, pooled_at=None):
    """(model, oracle layer list) of a TDNN stack: spec rows (units, context, padding, subsampling, form), form in affine / affine+relu /
    affine+relu+bn / own_<activation>; a reducing StatsPooling behind layer `pooled_at`."""
    lcfg = [{"name": "input", "type": "input", "shape": [None, None, D]}]
    for i, (U, ctx, pad, sub, form) in enumerate(spec):
        kinds = {"affine+relu+bn": ["affine", "relu", "batchnorm"], "affine+relu": ["affine", "relu"]}.get(form, "affine")
        c = {"units": U, "context": ctx, "padding": pad, "subsampling_factor": sub}
        if form.startswith("own_"):
            c["activation"] = form[4:]
        lcfg.append({"name": f"t{i}", "type": kinds, "cfg": c})
        if pooled_at == i + 1:
            lcfg.append({"name": "stats", "type": "stats", "cfg": {"left_context": 0, "right_context": 10000, "reduce_time_axis": True, "include_std": True}})
    mdl = ktf.models.SequentialFromConfig({"type": "sequential", "layers": lcfg}, None, "m", gemm=gemm)
    layers, din = [], D
    for i, (U, ctx, pad, sub, form) in enumerate(spec):
        W = (rng.standard_normal((U, len(ctx) * din)) / np.sqrt(len(ctx) * din)).astype(np.float32)
        b = (rng.standard_normal(U) * 0.1).astype(np.float32)
        mdl.get_layer(f"t{i}.affine").set_weights([W, b])
        lay = {"kind": "tdnn", "W": W, "b": b, "context": ctx, "padding": pad, "subsampling_factor": sub}
        if form.startswith("own_"):
            lay["activation"] = form[4:]
        layers.append(lay)
        if "relu" in form and not form.startswith("own_"):
            layers.append({"kind": "relu"})
        if form.endswith("bn"):
            bn = (np.float32(1.0), rng.uniform(-0.2, 0.4, U).astype(np.float32), rng.uniform(0.5, 2.0, U).astype(np.float32))
            mdl.get_layer(f"t{i}.batchnorm").set_weights(list(bn))
            layers.append({"kind": "bn", "rms": bn[0], "mean": bn[1], "var": bn[2]})
        din = U
        if pooled_at == i + 1:
            layers.append({"kind": "stats", "left_context": 0, "right_context": 10000, "reduce_time_axis": True, "include_std": True})
            din = 2 * U
    return mdl, layers


def test_flat_row_tiles_follow_the_lengths_a_valid_padded_layer_changed():
    """bf16x3 on flat row tiles (ktf_tdnn_split_flat) behind a VALID-padded, subsampling layer: the prefix sums of the lengths were made
    once per call, so a second flat layer behind such a layer tiled with the lengths of the first (found by tools/fuzz_models.py)."""
    rng = np.random.default_rng(5)
    spec = [(16, [-4, 4], "SAME", 1, "own_relu"), (130, [-1], "SAME", 1, "own_relu"), (96, [-1, 3], "VALID", 3, "affine+relu+bn"),
            (130, [-2], "SAME", 1, "affine")]
    mdl, layers = _stack(spec, 24, "bf16x3", rng)
    B, T = 40, 300
    lens = rng.integers(T // 3, T + 1, B).astype(np.int32)
    lens[0] = T
    x = rng.standard_normal((B, T, 24)).astype(np.float32)
    seen = []
    real = ops.tdnn_split_flat
    ops.tdnn_split_flat = lambda *a, **k: (seen.append(1), real(*a, **k))[1]
    try:
        got = mdl.run_ragged(dev(x), dev(lens)).cpu().numpy()
    finally:
        ops.tdnn_split_flat = real
    assert len(seen) == 2, "both 130-unit layers run on flat row tiles"
    for b in range(B):
        want = O.sequential_forward(layers, x[b:b + 1, : lens[b]], dtype=np.float64)[0]
        assert np.abs(got[b, : want.shape[0]] - want).max() < 3e-4 * np.abs(want).max(), b


@pytest.mark.parametrize("gemm", ["f32", "bf16", "f16mx"])
def test_utterances_that_lose_every_frame_give_nan_like_the_reference(gemm):
    """VALID padding can leave an utterance (or the whole batch) without a frame in front of the pooling: its mean is 0 / 0 and the NaN
    travels through the layers behind the pooling -- tf.nn.relu propagates it -- instead of turning into zeros or a null-pointer launch."""
    rng = np.random.default_rng(6)
    spec = [(300, [-2, 3], "VALID", 1, "affine+relu"), (256, [-4], "VALID", 1, "own_tanh"), (130, [-3, 4], "SAME", 2, "affine+relu+bn"),
            (130, [0], "SAME", 1, "affine"), (512, [0], "SAME", 1, "own_relu")]
    mdl, layers = _stack(spec, 40, gemm, rng, pooled_at=3)
    tol = {"f32": 2e-5, "bf16": 1.5e-1, "f16mx": 4e-3}[gemm]
    for lens in ([8, 11, 12, 12, 4], [5, 9, 8]):                   # some utterances / every utterance without a frame behind layer 2
        T = 12
        lens = np.asarray(lens, np.int32)
        x = rng.standard_normal((len(lens), T, 40)).astype(np.float32)
        got = mdl.run_ragged(dev(x), dev(lens)).float().cpu().numpy()
        for b in range(len(lens)):
            want = O.sequential_forward(layers, x[b:b + 1, : lens[b]], dtype=np.float64)[0]
            assert np.array_equal(np.isnan(got[b, :1]), np.isnan(want)), (gemm, lens.tolist(), b)
            if not np.isnan(want).any():
                assert np.abs(got[b, :1] - want).max() < tol * np.abs(want).max()
            else:
                assert np.isnan(want).all() and lens[b] < 10


def test_differential_fuzzing_of_the_sequential_runner_finds_nothing():
    """tools/fuzz_models.py: random TDNN stacks on ragged batches in every arithmetic mode against the oracle; a fixed seed here."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_models.py"), "60", "21"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "60 rounds, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # ... and with random settings of the runner's A/B knobs (kernel choices, flat row tiles, fused pooling): the seeds that found round 5's
    # three discrepancies (empty planes / tensors behind a VALID-padded layer that keeps no row; stale flat rows of a dense batch)
    for seed in ("606", "99", "202"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_models.py"), "120", seed, "--knobs"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "120 rounds, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # ... and the seed that found the finalize of the fused pooling turning the NaN deviation of an utterance without a frame into sqrt(eps)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_models.py"), "150", "9102"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "150 rounds, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_differential_fuzzing_of_the_extractor_finds_nothing():
    """tools/fuzz_extractor.py: batches mixing long, short, partly silent and completely silent utterances (NaN x-vectors, as in the
    reference), fp32 / int16 input, eager and captured, f32 / bf16x3 / f16mx with the shipped routing: every x-vector <= 1e-4 from the
    fp64 oracle; a fixed seed here."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_extractor.py"), "16", "31"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "16 rounds, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # ... with random settings of the runner's A/B knobs (kernel choices, fusion, routing) per round: they change schedules, not the contract
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_extractor.py"), "12", "33", "--knobs"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "12 rounds, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # ... and with a random front-end per round (sampling rate, frame sizes, mel / cepstrum options, VAD, CMVN: the generic front-end kernel)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_extractor.py"), "16", "32", "--cfg"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "16 rounds, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_captured_graphs_replayed_on_other_inputs_equal_eager_calls():
    """tools/fuzz_graph_replay.py: a graph captured on stationary noise, replayed on batches mixing speech, bursts + silence, quiet noise and
    digital silence (voiced lengths and the per-utterance routing change from replay to replay): bit-identical to the eager call."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_graph_replay.py"), "3", "41"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "3 replays per graph, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_models_on_concurrent_host_threads_and_streams():
    """tools/thread_probe.py: four host threads, each with its own extractor (f16mx / bf16x3 / f32) and HIP stream, 20 calls each: every
    x-vector equals the single-threaded result bit for bit."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "thread_probe.py"), "20"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "equals the single-threaded result" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
