"""Where the headline mode's parity margin is thin (run with `-m gpu` on an MI355X): `f16mx` on SHORT speech windows and on
weights whose BatchNorm statistics are the network's own.

The block-scaled residual terms of `f16mx` leave zero-mean rounding noise per frame that the statistics pooling averages over
the voiced frames, so the x-vector deviation grows as the pooled window shrinks: 10 s 2-5e-5, 5 s up to 6e-5, 1 s up to 9e-5 of
the 1e-4 tolerance (fp64 emulation: tools/emulate_schemes.py `mx4_46`). The diarization recipes the reference documents use
1.5 s windows (README.md:203-206 of the reference), so that regime is gated here on the reference's own
speech recording (models/kaldi/xvector_extractor_test.py:70-96, tests/golden/e2e_0008.npz:wav_int16) at a 1 s hop, one batch per
window length, four weight seeds, against the fp64 oracle:

* the RAW kernels (routing off): split-bf16 stays inside 2e-5 at every window length; `f16mx` is inside 1e-4 from 3 s on and
  OUTSIDE it below (measured 7-12e-5 at 1 s and 1.5 s, 4-7.5e-5 at 3 s, 4-6.5e-5 at 5 s; the self-consistent BatchNorm weights are
  the worse case): bounded here at 1.5e-4 so that a regression shows, and the reason for the routing;
* the SHIPPED routing (`Sequential.MIN_FRAMES`, 400 frames): shorter windows go to the split-bf16 kernels -- whole batches by their
  frame count on the host, single utterances of a long batch by their voiced-frame count on the device
  (`XvectorExtractor.route_short_utterances`) -- and what still reaches the `f16mx` kernels stays inside 8e-5;
* the same on SELF-CONSISTENT BatchNorm weights (tests/_selfbn.py): the closest offline stand-in for trained statistics.
"""

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import ops
from oracle import ktf_oracle as O
from _selfbn import self_consistent_weights

pytestmark = pytest.mark.gpu

TOL = 1e-4                  # north_star: max-abs x-vector deviation
ROUTED_BOUND = 8e-5         # what the shipped routing guarantees on the gated regimes
SEEDS = [4321, 1, 2, 3]
WINDOWS_S = [1.0, 1.5, 3.0, 4.1, 5.0]          # 98, 148, 298, 408, 498 frames


def windows(sec, hop=1.0):
    """(n, samples) windows of the reference's speech recording, `hop` seconds apart."""
    sp = synth.speech_wavs()[0][0]
    n, h = int(sec * 16000), int(hop * 16000)
    return np.stack([sp[s:s + n] for s in range(0, len(sp) - n + 1, h)], 0)


_weights, _want = {}, {}


def weights(kind, seed):
    if (kind, seed) not in _weights:
        if kind == "synthetic":
            _weights[(kind, seed)] = synth.make_weights(seed=seed)
        else:       # BatchNorm statistics = the network's own statistics on the recording itself
            _weights[(kind, seed)] = self_consistent_weights(seed, [synth.speech_wavs()[0][0]])
    return _weights[(kind, seed)]


def oracle(kind, seed, sec):
    if (kind, seed, sec) not in _want:
        w = weights(kind, seed)
        _want[(kind, seed, sec)] = O.xvector_forward(windows(sec), synth.extractor_cfg(), synth.oracle_layers(w), w["mean"], w["lda"],
                                                     dtype=np.float64)
    return _want[(kind, seed, sec)]


def deviations(gemm, kind, seed, routed):
    """{window seconds: (max-abs deviation over the batch of windows, kernel family of the FIRST frame-level GEMM launch)}"""
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), weights(kind, seed), gemm=gemm)
    mdl.xvec.min_tiles = {}                               # a batch of ~20 windows is a handful of tiles: keep it off the fp32 small-tile route
    if not routed:
        mdl.xvec.min_frames = {}
    out = {}
    names = ("tdnn", "tdnn_split", "tdnn_split_flat", "tdnn_mx")
    orig = {n: getattr(ops, n) for n in names}
    for sec in WINDOWS_S:
        wav = windows(sec)
        first = []

        def wrap(fn):
            def f(*a, **k):
                r = fn(*a, **k)
                if not first:
                    first.append(ops.last_kernel())
                return r
            return f
        for n in names:
            setattr(ops, n, wrap(orig[n]))
        try:
            got = mdl(torch.as_tensor(wav, device="cuda")).cpu().numpy().reshape(wav.shape[0], -1)
        finally:
            for n in names:
                setattr(ops, n, orig[n])
        out[sec] = (float(np.abs(got - oracle(kind, seed, sec)).max()), first[0])
    return out


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("kind", ["synthetic", "self_consistent_bn"])
@pytest.mark.parametrize("gemm", ["bf16x3", "f16mx"])
def test_raw_kernels_on_short_speech_windows(gemm, kind, seed):
    """Routing off: the mode's own kernels at every window length. split-bf16: inside 2e-5 everywhere. f16mx: inside the
    tolerance from 3 s on; on 1 s / 1.5 s windows its block-scaled rounding noise is averaged over too few frames (up to 1.2e-4
    measured at 1 s): bounded at 1.5e-4, and the shipped routing below keeps such batches off it."""
    d = deviations(gemm, kind, seed, routed=False)
    print(f"{gemm} raw, {kind} weights, seed {seed}: " + ", ".join(f"{k:g} s {v[0]:.2e}" for k, v in d.items()))
    if gemm == "bf16x3":
        assert max(v[0] for v in d.values()) <= 2e-5, d
        return
    for sec, (err, _) in d.items():
        frames = 1 + (int(sec * 16000) - 400) // 160
        assert err <= (TOL if frames >= 290 else 1.5e-4), (sec, err)


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("kind", ["synthetic", "self_consistent_bn"])
def test_shipped_routing_on_short_speech_windows(kind, seed):
    """`Sequential.MIN_FRAMES` as shipped: 1 s, 1.5 s and 3 s windows (98 / 148 / 298 frames) run the split-bf16 kernels, 4.1 s and 5 s
    windows the f16mx kernels, and every gated regime stays inside 8e-5."""
    d = deviations("f16mx", kind, seed, routed=True)
    print(f"f16mx routed, {kind} weights, seed {seed}: " + ", ".join(f"{k:g} s {v[0]:.2e} ({v[1]})" for k, v in d.items()))
    for sec, (err, kernel) in d.items():
        short = ktf.models.Sequential.MIN_FRAMES["f16mx"] > 1 + (int(sec * 16000) - 400) // 160
        assert ("x3" in kernel) == short, (sec, kernel)
        assert ("mx" in kernel) == (not short), (sec, kernel)
        assert err <= ROUTED_BOUND, (sec, err, kernel)


def test_min_frames_is_a_per_model_knob():
    """The routing knobs are instance attributes copied from the class defaults: changing one model's does not move another's."""
    w = weights("synthetic", 4321)
    a = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm="f16mx")
    b = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm="f16mx")
    a.xvec.min_frames, a.xvec.min_tiles = {}, {}
    assert b.xvec.min_frames == ktf.models.Sequential.MIN_FRAMES and b.xvec.min_tiles == ktf.models.Sequential.MIN_TILES
    assert a.xvec.batch_gemm(64, 148) == ktf._lib.GEMM_F16MX
    assert b.xvec.batch_gemm(64, 148) == ktf._lib.GEMM_BF16X3 and b.xvec.batch_gemm(64, 398) == ktf._lib.GEMM_BF16X3
    assert b.xvec.batch_gemm(64, 998) == ktf._lib.GEMM_F16MX
    assert b.xvec.batch_gemm(1, 998) == ktf._lib.GEMM_F32


# ----------------------------------------------------------------------------- per-utterance routing on the device
def _embedded(sec_voiced, total=160000, seed=0, where=20000):
    """`sec_voiced` seconds of the speech recording inside `total` samples of noise at 1e-3 gain (the VAD drops it)."""
    rng = np.random.default_rng(seed)
    x = np.round(rng.standard_normal(total)).astype(np.float32)          # sigma 1: far below the energy threshold
    sp = synth.speech_wavs()[0][0]
    n = int(sec_voiced * 16000)
    x[where:where + n] = sp[30000 + 7000 * seed:30000 + 7000 * seed + n]
    return x


@pytest.mark.parametrize("seed", SEEDS[:2])
def test_vad_shortened_utterances_are_routed_on_the_device(seed):
    """10 s recordings of which the VAD keeps 1.2 - 1.8 s: the batch is long (998 frames: the host-side MIN_FRAMES rule does not fire)
    but these utterances pool over fewer than 400 frames. XvectorExtractor.route_short_utterances (default) sends exactly them
    through the split-bf16 kernels, decided per utterance on the device from the voiced-frame counts: every x-vector inside 8e-5,
    the long utterances bit-identical to an unrouted run, an utterance without a voiced frame still NaN."""
    cfg = synth.extractor_cfg()
    w = weights("synthetic", seed)
    sp = synth.speech_wavs()[0][0]
    wav = np.stack([sp[:160000], _embedded(1.2, seed=1), sp[160000:320000], _embedded(1.8, seed=2), _embedded(1.5, seed=3),
                    np.round(np.random.default_rng(5).standard_normal(160000)).astype(np.float32)], 0)
    mdl = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
    mdl.xvec.min_tiles = {}
    y = mdl(torch.as_tensor(wav, device="cuda"))
    assert mdl.last_short_count == 3
    lens = mdl.last_lens.cpu().numpy()
    assert lens[0] > 400 and lens[2] > 400 and lens[5] == 0 and all(0 < lens[i] < 400 for i in (1, 3, 4)), lens
    got = y.cpu().numpy()
    assert np.isnan(got[5]).all()
    want = O.xvector_forward(wav[:5], cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    err = np.abs(got[:5] - want).max(1)
    mdl.route_short_utterances = False
    raw = mdl(torch.as_tensor(wav, device="cuda")).cpu().numpy()
    err_raw = np.abs(raw[:5] - want).max(1)
    print(f"seed {seed}: voiced frames {lens.tolist()}; routed {np.array2string(err, precision=2)}; unrouted {np.array2string(err_raw, precision=2)}")
    assert err.max() <= ROUTED_BOUND, err
    assert np.array_equal(got[[0, 2]], raw[[0, 2]]), "long utterances must not depend on the routing of the others"
    assert not np.array_equal(got[[1, 3, 4]], raw[[1, 3, 4]])
    # the same through a captured graph (the routing is device-side: nothing to decide on the host)
    mdl.route_short_utterances = True
    run = mdl.compile(torch.as_tensor(wav, device="cuda"))
    assert torch.equal(torch.nan_to_num(run(torch.as_tensor(wav, device="cuda"))), torch.nan_to_num(y))


def test_device_routing_on_the_large_batch_tail_route():
    """From 512 utterances on the first pass ends in the GEMM tail (tdnn6 over the batch) while the second pass uses the fused tail
    with KTF_TAIL_SKIP_EMPTY: rows of long utterances keep the first pass's bits, short ones get the split-bf16 result."""
    cfg = synth.extractor_cfg()
    w = weights("synthetic", 4321)
    rng = np.random.default_rng(3)
    n = 400 + 160 * 430                                     # 431 frames
    wav = synth.make_wav(512, n, seed=77)
    short = rng.random(512) < 0.1
    wav[short, 12000:] = np.round(wav[short, 12000:] * 1e-3)         # ~73 voiced frames left
    mdl = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
    y = mdl(torch.as_tensor(wav, device="cuda"))
    lens = mdl.last_lens.cpu().numpy()
    assert (lens[short] < 400).all() and (lens[~short] >= 400).all()
    assert mdl.last_short_count == int(short.sum())
    assert mdl(torch.as_tensor(wav[~short], device="cuda")).shape[0] == int((~short).sum()) and mdl.last_short_count == 0       # nothing short: one pass
    ref = synth.build_extractor(ktf, cfg, w, gemm="bf16x3")
    ref.route_short_utterances = False
    yb = ref(torch.as_tensor(wav, device="cuda"))
    mdl.route_short_utterances = False
    yr = mdl(torch.as_tensor(wav, device="cuda"))
    s = torch.as_tensor(short, device="cuda")
    assert torch.equal(y[~s], yr[~s])
    assert (y[s] - yb[s]).abs().max().item() <= 2e-5 and not torch.equal(y[s], yr[s])


# ------------------------------------------------------------------------------------------------ round 5: the thin spot, widened
# the 1e-4 claim of the timed mode rested on four weight seeds, one recording and BatchNorm statistics of one family, and
# the routing threshold (MIN_FRAMES = 400) was tuned on the same data. Here: 16 weight draws of three families -- the synthetic
# default, BatchNorm variances over two orders of magnitude (U(0.05, 4): units far from unit scale), heavy-tailed weight rows
# (Student t, 3 degrees of freedom) --, every voiced length from 400 to 600 frames in steps of 20 (just above the threshold: where the
# block-scaled rounding noise is averaged over the fewest frames), windows of both recordings the reference ships (the 22.5 s
# end-to-end input and its 3 s feature-test clip, looped), and three input gains (the MFCC / CMVN front-end is gain-invariant up to the
# VAD's absolute energy threshold and the dither-free log floor). Everything through the SHIPPED routing, against the fp64 oracle.
SWEEP_FRAMES = list(range(400, 601, 20))
SWEEP_FRAMES_OUTLIERS = [640, 680, 720, 760]          # ... and for the weight families the model routes below 640 frames (frames_floor)
SWEEP_FAMILIES = [("synthetic", {}), ("wide_bn", {"bn": "wide"}), ("t3_rows", {"tails": "t3"}), ("wide_bn_t3_rows", {"bn": "wide", "tails": "t3"})]
SWEEP_BOUND = 8e-5


def sweep_windows(frames):
    """(8, samples) windows of `frames` frames: seven of the long recording 2.5 s apart, one of the 3 s clip looped to length."""
    n = (frames - 1) * 160 + 400
    sp = synth.speech_wavs()[0][0]
    clip = synth.second_speech_wav()
    rows = [sp[s:s + n] for s in range(0, len(sp) - n + 1, 40000)][:7]
    rows.append(np.tile(clip, 1 + n // len(clip))[:n])
    return np.stack(rows, 0)


@pytest.mark.parametrize("seed", list(range(16)))
def test_f16mx_margin_sweep_400_to_600_frames(seed):
    """One weight draw, 11 window lengths x 8 windows: the shipped f16mx model stays inside 8e-5 of the fp64 oracle on every one
    (tolerance 1e-4). Seeds 0-3 also run at input gains 0.01 and 30. The Student-t families hold weights 30-150 standard deviations out
    (seed 1014: one window of 520 frames at 8.9e-5 on the f16mx kernels): `Sequential.frames_floor` sees them and routes below 640
    frames, so for those families the sweep goes on to 760 frames, where the f16mx kernels take over."""
    name, kw = SWEEP_FAMILIES[seed % 4]
    w = synth.make_weights(seed=1000 + seed, **kw)
    cfg = synth.extractor_cfg()
    mdl = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
    mdl.xvec.min_tiles = {}                               # (eight windows are a handful of tiles: keep them off the small-tile route)
    outliers = "tails" in kw
    assert (mdl.xvec.weight_outlier_score() > mdl.xvec.OUTLIER_SIGMAS) == outliers
    assert mdl.xvec.frames_floor("f16mx") == (640 if outliers else 400)
    layers = synth.oracle_layers(w)
    worst = {}
    for gain in ([0.01, 1.0, 30.0] if seed < 4 else [1.0]):
        for frames in SWEEP_FRAMES + (SWEEP_FRAMES_OUTLIERS if outliers else []):
            wav = (sweep_windows(frames) * gain).astype(np.float32)
            got = mdl(torch.as_tensor(wav, device="cuda")).cpu().numpy().reshape(wav.shape[0], -1)
            want = O.xvector_forward(wav, cfg, layers, w["mean"], w["lda"], dtype=np.float64)
            ok = np.isfinite(want).all(1)                 # (a window the VAD empties is NaN in both: the reference's 0 / 0)
            assert np.array_equal(ok, np.isfinite(got).all(1)), (gain, frames)
            if ok.any():
                worst[(gain, frames)] = float(np.abs(got[ok] - want[ok]).max())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"{name} weights, seed {1000 + seed}: max {top[0][1]:.2e} at (gain, frames) {top[0][0]}; next {top[1][1]:.2e} {top[1][0]}, {top[2][1]:.2e} {top[2][0]}")
    assert top[0][1] <= SWEEP_BOUND, top


def test_verify_fraction_reports_what_the_mode_costs_on_this_model():
    """XvectorExtractor.verify_fraction: a random part of every batch is extracted again on the tighter kernels and the largest
    difference lands in `last_verify` -- a deployment on weights nobody tested sees its own margin. On the synthetic weights it agrees with
    what the oracle says about the same rows to ~1e-5 (the tighter mode's own distance from fp64)."""
    w = weights("synthetic", 4321)
    cfg = synth.extractor_cfg()
    mdl = synth.build_extractor(ktf, cfg, w, gemm="f16mx")
    mdl.xvec.min_tiles = {}
    wav = windows(5.0)                                    # 18 windows of 498 frames: f16mx kernels
    assert mdl.last_verify is None
    y0 = mdl(torch.as_tensor(wav, device="cuda")).cpu().numpy()
    assert mdl.last_verify is None                        # off by default
    mdl.verify_fraction = 0.25
    y1 = mdl(torch.as_tensor(wav, device="cuda")).cpu().numpy()
    v = mdl.last_verify
    assert np.array_equal(y0, y1)                         # the guard does not touch the result
    assert v["n"] == 5 and len(v["rows"]) == 5 and v["mode"] == "bf16x3" and 0 < v["max_abs_dev"] <= TOL
    dev_oracle = np.abs(y1[v["rows"]] - oracle("synthetic", 4321, 5.0)[v["rows"]]).max()
    assert abs(v["max_abs_dev"] - dev_oracle) <= 2.5e-5, (v, dev_oracle)
    mdl(torch.as_tensor(wav, device="cuda"))
    assert mdl.last_verify["running_max"] >= v["max_abs_dev"]
