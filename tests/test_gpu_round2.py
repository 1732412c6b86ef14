"""GPU parity tests added in round 2 (run with `-m gpu` on an MI355X): the Kaldi golden that pins a12, the whole
pipeline at the FULL BASELINE size against the fp64 oracle, the chained config-5 path (wav -> x-vector -> PLDA 1024 x 1024),
degenerate VAD lengths, recordings longer than the fused kernel's LDS map, reproducible fused pooling, the hipGraph
product path, the bounded workspace, and the Windowing layer on the reference test's 1000 frames x 11 overrides."""

import json

import numpy as np
import pytest
import torch

import _golden as G
import synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O

pytestmark = pytest.mark.gpu
Ls = ktf.layers


@pytest.fixture(autouse=True, scope="module")
def _reduced_modes_reach_their_kernels():
    S = ktf.models.Sequential           # defaults of models built in this module (instances copy them; no call-time global)
    old = (S.MIN_TILES, S.MIN_FRAMES)
    S.MIN_TILES, S.MIN_FRAMES = {}, {}
    yield
    S.MIN_TILES, S.MIN_FRAMES = old


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda").to(dtype)


def host(t):
    return t.detach().to(torch.float64 if t.dtype == torch.float64 else torch.float32).cpu().numpy()


# ----------------------------------------------------------------------------- a12 pinned to Kaldi
def test_xvec_post_kaldi_golden():
    """mean-subtraction + LDA + length-norm (xvector_extractor.py:174-184) against Kaldi's own output for the same 512-d
    embedding (testdata/models/src/0008_sitw_v2_1a/xvector.unnorm.ark.txt -> xvector.ark.txt, compute_xvectors.sh)."""
    z = G.load("e2e_0008.npz")
    mean = ktf.io.ReadKaldiArray(G.GOLDEN + "/xvectors_train_combined_200k.mean.vec.txt", binary=False)
    lda = ktf.io.ReadKaldiArray(G.GOLDEN + "/xvectors_train_combined_200k.transform.mat", binary=True)
    x = dev(z["xvector_unnorm"].reshape(1, 512))
    A = dev(np.ascontiguousarray(lda[:, :-1].T))
    got = host(ktf.ops.xvec_post(x, dev(mean), A, dev(lda[:, -1].copy())))
    assert got.shape == (1, 128)
    assert np.abs(got - z["xvector"].reshape(1, 128)).max() <= 1e-5
    # the same through the model object (from_parts routes the real fixture files through _setup)
    w = synth.make_weights(seed=1, narrow=True)
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), w)
    m, a, o = (ktf.ops.to_device_f32(v) for v in (mdl.xvecGlobalMean, mdl.ldaMat, mdl.ldaOffset.reshape(-1)))
    assert np.abs(host(ktf.ops.xvec_post(x, m, a, o)) - z["xvector"].reshape(1, 128)).max() <= 1e-5


# ----------------------------------------------------------------------------- whole pipeline at the BASELINE size
@pytest.mark.parametrize("gemm", ["f32", "bf16x3", "f16mx"])
def test_extractor_full_topology_10s_vs_oracle(gemm):
    """0008 topology, full widths, 160 000-sample utterances (998 frames): max-abs deviation from the fp64 oracle within
    the north_star bound (1e-4), for the exact path, the split-bf16 path and the block-scaled path bench.py times by default.
    One utterance is all-voiced stationary noise (the bench workload), two have quiet blocks (ragged)."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321, narrow=False)
    wav = np.concatenate([synth.make_wav(1, 160000, seed=1234), synth.make_wav(2, 160000, seed=4242, ragged=True)], 0)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    mdl = synth.build_extractor(ktf, cfg, w, gemm=gemm)
    got = host(mdl(dev(wav)))
    lens = mdl.last_lens.cpu().numpy()
    assert lens[0] == 998 and (lens[1:] < 998).all()
    err = np.abs(got - want).max()
    print(f"extractor {gemm}, 10 s, full topology: max-abs dev vs fp64 oracle {err:.3e}")
    assert err <= 1e-4, (gemm, err)


def test_config5_wav_to_xvector_to_plda_1024():
    """BASELINE config 5 end to end: 1024 utterances -> x-vectors -> PLDA 1024 x 1024 trial matrix. The x-vectors of a
    sample of utterances are checked against the fp64 oracle, and the scores of those trials against the oracle's PLDA on
    the oracle's x-vectors (chained tolerance), plus the whole matrix against the oracle's PLDA on the GPU x-vectors."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321, narrow=False)
    B, N = 1024, 32000
    wav = synth.make_wav(B, N, seed=2025, ragged=True)
    mdl = synth.build_extractor(ktf, cfg, w, gemm="f32")
    xv = mdl(dev(wav))
    assert tuple(xv.shape) == (B, 128) and bool(torch.isfinite(xv).all())
    rng = np.random.default_rng(31)
    dim = 128
    A = rng.standard_normal((dim, dim)) / np.sqrt(dim) + np.eye(dim)
    mean = rng.standard_normal(dim) * 0.1
    psi = np.sort(rng.uniform(0.05, 30.0, dim))[::-1].copy()
    plda = Ls.PLDA(dim, mean, A, psi)
    scores, _ = plda(xv)
    s = host(scores)
    assert s.shape == (B, B)
    full, _ = O.plda(host(xv).astype(np.float64), mean, A, psi, dtype=np.float64)
    assert np.abs(s - full).max() < 1e-7 * max(1.0, np.abs(full).max())
    pick = [0, 1, 2, 511, 1023]
    want_x = O.xvector_forward(wav[pick], cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    assert np.abs(host(xv)[pick] - want_x).max() <= 1e-4
    want_s, _ = O.plda(want_x, mean, A, psi, dtype=np.float64)
    got_s = s[np.ix_(pick, pick)]
    assert np.abs(got_s - want_s).max() < 2e-2 and np.abs(got_s - want_s).max() < 1e-3 * np.abs(want_s).max()


# ----------------------------------------------------------------------------- VAD on fewer than 2*context frames
def test_vad_degenerate_lengths_follow_the_reference_scatter():
    """vad.py:187-193 scatters the edge window sizes at indexes modulo T; with T < 2*frames_context the writes collide and
    the last one stands (oracle = that order). T = 1 .. 6 with context 2 and 3, mask and index forms, and the fused
    VAD + CMVN kernel's kept-frame count."""
    rng = np.random.default_rng(5)
    for ctx in (2, 3):
        for T in range(1, 2 * ctx + 3):
            for trial in range(6):
                feats = (rng.standard_normal((3, T, 30)) * 4 + 6).astype(np.float32)
                for prop in (0.12, 0.3, 0.6):
                    cfg = dict(energy_mean_scale=0.5, energy_threshold=5.5, frames_context=ctx, proportion_threshold=prop)
                    mask = host(Ls.VAD(**cfg, return_indexes=False)(feats))
                    want = O.vad(feats, **cfg, return_indexes=False)
                    assert np.array_equal(mask, want), (ctx, T, trial, prop)
                    idx = Ls.VAD(**cfg, return_indexes=True)(feats).cpu().numpy()
                    assert np.array_equal(idx, O.vad(feats, **cfg, return_indexes=True)), (ctx, T, trial, prop)
    # fused kernel: lens = number of kept frames
    feats = (rng.standard_normal((4, 3, 30)) * 4 + 6).astype(np.float32)
    cfg = dict(energy_mean_scale=0.5, energy_threshold=5.5, frames_context=2, proportion_threshold=0.12)
    out = torch.zeros((4, 3, 32), device="cuda")
    lens = torch.zeros((4,), dtype=torch.int32, device="cuda")
    idx = torch.zeros((4, 3), dtype=torch.int32, device="cuda")
    work = torch.zeros((4 * 3 * 60 + 60,), device="cuda")
    ktf.ops.vad_cmvn(dev(feats), Ls.VAD(**cfg).cfg(), Ls.CMVN(window=300).cfg(), out, lens, idx, work)
    want = O.vad(feats, **cfg, return_indexes=False)[..., 0].sum(-1)
    assert np.array_equal(lens.cpu().numpy(), want.astype(np.int32))


# ----------------------------------------------------------------------------- recordings beyond the LDS frame map
def test_long_recording_beyond_38400_frames():
    """A 6 min 50 s recording (40 998 frames): the frame -> row map of the fused VAD/CMVN kernel no longer fits in LDS and
    lives in the caller's idx_work; results still match the fp64 oracle (narrow network: the oracle runs in seconds)."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=1, narrow=True)
    mdl = synth.build_extractor(ktf, cfg, w)
    n = 400 + 160 * 40997
    wav = synth.make_wav(2, n, seed=314, ragged=True)
    want, inter = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64,
                                    return_intermediates=True)
    mfcc, feats, lens = mdl.features(dev(wav))
    assert mfcc.shape[1] == 40998
    lens_h = lens.cpu().numpy()
    for b in range(2):
        assert lens_h[b] == len(inter[b]["voiced"]) and lens_h[b] < 40998
        assert np.abs(host(feats[b, : lens_h[b]]) - inter[b]["cmvn"]).max() < 2e-3
    assert np.abs(host(mdl(dev(wav))) - want).max() <= 1e-4


# ----------------------------------------------------------------------------- reproducible fused pooling
@pytest.mark.parametrize("gemm", ["bf16x3", "f16mx", "bf16"])
def test_fused_pooling_is_reproducible_and_matches_the_atomic_form(gemm):
    """KTF_TDNN_DET_STATS (the models' default): per-128-row partial sums added in block order -> bitwise identical
    x-vectors run after run and batch == single; the fp64-atomic form agrees to the last fp32 bits."""
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321, narrow=False)
    wav = dev(synth.make_wav(24, 16000 * 4 + 333, seed=77, ragged=True))
    det = synth.build_extractor(ktf, cfg, w, gemm=gemm)
    assert det.xvec.deterministic
    a = det(wav)
    for _ in range(3):
        assert torch.equal(det(wav), a)
    if gemm in ("bf16x3", "f16mx"):   # utterances that fill their tiles badly run on flat row tiles: the pooled layer's partial sums follow the flat row space (Sequential.flat_pooling)
        for i in (0, 7, 23):
            assert float((det(wav[i:i + 1]) - a[i]).abs().max()) <= 2e-6
        det.xvec.flat_pooling = False
        a = det(wav)
    for i in (0, 7, 23):
        assert torch.equal(det(wav[i:i + 1]), a[i])
    atom = synth.build_extractor(ktf, cfg, w, gemm=gemm)
    atom.xvec.deterministic = False
    b = atom(wav)
    assert float((a - b).abs().max()) < 2e-5


# ----------------------------------------------------------------------------- hipGraph product path
@pytest.mark.parametrize("gemm,B", [("f32", 1), ("bf16x3", 16), ("f16mx", 16)])
def test_compiled_extractor_replays_bitwise(gemm, B):
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=4321, narrow=False)
    mdl = synth.build_extractor(ktf, cfg, w, gemm=gemm)
    x0 = dev(synth.make_wav(B, 160000, seed=3, ragged=True))
    run = mdl.compile(x0)
    for seed in (3, 4, 5):
        x = dev(synth.make_wav(B, 160000, seed=seed, ragged=True))
        want = mdl(x)
        got = run(x)
        assert got.shape == want.shape and torch.equal(got, want), (gemm, seed)
    assert torch.equal(run(x0.to(torch.float32)), mdl(x0))
    with pytest.raises(ValueError):
        run(x0[:, :-160])


# ----------------------------------------------------------------------------- workspace stays bounded
def test_workspace_is_bounded_by_the_largest_request():
    cfg = synth.extractor_cfg()
    w = synth.make_weights(seed=1, narrow=True)
    mdl = synth.build_extractor(ktf, cfg, w)
    big = dev(synth.make_wav(4, 16000 * 5, seed=1, ragged=True))
    ref_big = mdl(big).clone()
    cap = mdl._ws.bytes() + mdl.xvec._ws.bytes()
    outs = {}
    for i, n in enumerate([16000, 20000 + 37, 31000, 47000, 16000 * 5 - 160, 8000]):
        x = dev(synth.make_wav(1 + i % 4, n, seed=10 + i, ragged=True))
        outs[n] = (x, mdl(x).clone())
    assert mdl._ws.bytes() + mdl.xvec._ws.bytes() == cap                  # no growth for smaller shapes
    assert torch.equal(mdl(big), ref_big)                                 # re-viewed arenas give the same bits
    for n, (x, y) in outs.items():
        assert torch.equal(mdl(x), y), n
    # a second stream gets arenas of its own (no sharing of scratch across streams)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y2 = mdl(big)
    torch.cuda.current_stream().wait_stream(s)
    assert torch.equal(y2, ref_big) and mdl._ws.bytes() + mdl.xvec._ws.bytes() > cap


# ----------------------------------------------------------------------------- Windowing at the reference test's size
def test_windowing_1000_frames_11_overrides():
    """layers/dsp/windowing_test.py:85-120: 1000 random frames x 11 option overrides against ProcessFrames, tolerance
    2 * dither (RMSE 2e-7 where dither = 0). ktf.kaldi_numpy.ProcessFrames is itself pinned to the reference's NumPy
    outputs on the first 40 of these frames (same RandomState(12345) stream; tests/test_host_cpu.py)."""
    frames = np.random.RandomState(12345).random_sample((1, 1000, 256))
    z = G.load("kaldi_numpy.npz")
    assert np.array_equal(frames[:, :40], z["frames"])
    overrides = json.loads(str(z["configs_json"])) + [{"dither": 1.0}]
    assert len(overrides) == 11
    for o in overrides:
        cfg = {"window_type": "povey", "blackman_coeff": 0.42, "dither": 0.0, "remove_dc_offset": True,
               "preemphasis_coefficient": 0.97, "raw_energy": True, "return_energy": True, "energy_floor": 0.0,
               "epsilon": float(np.finfo(np.float32).eps)}
        cfg.update(o)
        np.random.seed(1)
        want_w, want_e = ktf.kaldi_numpy.ProcessFrames(frames, dither=cfg["dither"], remove_dc_offset=cfg["remove_dc_offset"],
                                                       preemphasis_coefficient=cfg["preemphasis_coefficient"],
                                                       window_type=cfg["window_type"], raw_energy=cfg["raw_energy"])
        w, e = Ls.Windowing(**cfg)(frames)
        assert tuple(w.shape) == frames.shape and tuple(e.shape) == (1, 1000, 1)
        tol = 2 * cfg["dither"] if cfg["dither"] else 2e-7
        assert G.rmse(want_w, host(w)) < tol and G.rmse(want_e, host(e)) < tol, o


@pytest.mark.gpu
def test_workspace_refills_only_views_with_pad_columns():
    """A role whose shape changes is re-zeroed only when the new view has pad columns nobody writes; in steady state the
    single-utterance path launches no fill kernel (three per call before: 7 % of its latency)."""
    import torch
    from kaldi_tflite_amd.models import _Workspace
    dev = torch.device("cuda", 0)
    ws = _Workspace()
    a = ws.get("act0", (4, 8), torch.float32, dev, padded=False)
    a.fill_(3.0)
    b = ws.get("act0", (2, 8), torch.float32, dev, padded=False)        # re-sliced, no pad columns: handed out as is
    assert float(b.sum()) == 48.0
    c = ws.get("act0", (2, 12), torch.float32, dev, padded=True)        # pad columns must read as zeros
    assert float(c.abs().sum()) == 0.0
    c.fill_(1.0)
    assert ws.get("act0", (2, 12), torch.float32, dev, padded=True) is c and float(c.sum()) == 24.0      # same shape: same view, untouched
    assert ws.bytes() == 4 * 8 * 4
