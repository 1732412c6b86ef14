"""GPU parity on NON-STATIONARY audio (run with `-m gpu` on an MI355X): every GEMM arithmetic mode that claims the north_star
tolerance (<= 1e-4 max-abs x-vector deviation) is checked on the reference's own end-to-end input -- 22.5 s of read speech
(testdata/librispeech_2.wav, models/kaldi/xvector_extractor_test.py:70-96; committed as tests/golden/e2e_0008.npz:wav_int16) --
whole and as 10 s chunks, plus amplitude-modulated coloured noise, over several weight seeds, against the fp64 oracle. The
stationary noise of the throughput workload cannot show errors that depend on the input distribution: a constant bias per
utterance survives the statistics pooling, zero-mean rounding noise does not (round 2's calibrated one-pass form passed on
noise and is 4-7e-4 on speech: it is kept below as a documented out-of-tolerance mode)."""

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O
from _selfbn import self_consistent_weights as _self_consistent_weights

pytestmark = pytest.mark.gpu

TOL = 1e-4
COMPLIANT = ["f32", "bf16x3"] + (["f16mx"] if "f16mx" in ktf.layers._GEMM else [])
SEEDS = [4321, 1, 2, 3]


@pytest.fixture(autouse=True, scope="module")
def _reduced_modes_reach_their_kernels():
    S = ktf.models.Sequential           # defaults of models built in this module (instances copy them; no call-time global)
    old = (S.MIN_TILES, S.MIN_FRAMES)
    S.MIN_TILES, S.MIN_FRAMES = {}, {}
    yield
    S.MIN_TILES, S.MIN_FRAMES = old


_inputs = {}


def inputs():
    if not _inputs:
        whole, chunks = synth.speech_wavs()
        _inputs.update({"speech_22s": whole, "speech_10s_chunks": chunks, "coloured_am_noise": synth.coloured_am_noise(2, 160000)})
    return _inputs


_want = {}


def oracle(seed, name):
    """fp64 oracle x-vectors of input `name` under weight seed `seed` (cached: shared by the modes)."""
    if (seed, name) not in _want:
        w = synth.make_weights(seed=seed)
        _want[(seed, name)] = O.xvector_forward(inputs()[name], synth.extractor_cfg(), synth.oracle_layers(w), w["mean"], w["lda"],
                                                dtype=np.float64)
    return _want[(seed, name)]


def deviations(gemm, seed, calibrate=False, prepare=None):
    w = synth.make_weights(seed=seed)
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm=gemm, calibrate=calibrate)
    if prepare is not None:
        prepare(mdl)
    out = {}
    for name, wav in inputs().items():
        got = mdl(torch.as_tensor(wav, device="cuda")).cpu().numpy().reshape(wav.shape[0], -1)
        out[name] = float(np.abs(got - oracle(seed, name)).max())
    return out


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("gemm", COMPLIANT)
def test_compliant_modes_on_speech(gemm, seed):
    """Modes that carry `tolerance_ok: true` in bench.py: <= 1e-4 on speech (whole and in 10 s chunks) and on modulated noise."""
    d = deviations(gemm, seed)
    print(f"{gemm} seed {seed}: " + ", ".join(f"{k} {v:.2e}" for k, v in d.items()))
    assert max(d.values()) <= TOL, (gemm, seed, d)


@pytest.mark.parametrize("seed", SEEDS[:2])
def test_f16x2_on_speech_is_outside_the_tolerance_when_calibrated_on_noise(seed):
    """The round-2 default (one-pass tail + residual prefix, calibrated on stationary noise) keeps delta_W (x_utt_mean - x_cal_mean)
    per utterance: measured 4-7e-4 on speech. It stays available as an explicit option and is reported with tolerance_ok false;
    this test pins the order of magnitude so that a silent change of the route shows up."""
    d = deviations("f16x2", seed, calibrate=True)
    print(f"f16x2 calibrated on noise, seed {seed}: " + ", ".join(f"{k} {v:.2e}" for k, v in d.items()))
    assert max(d.values()) <= 5e-3
    assert d["speech_10s_chunks"] > TOL or d["speech_22s"] > TOL, "the calibrated form is expected to miss the tolerance on speech"


@pytest.mark.parametrize("seed", SEEDS)
def test_f16x2_two_pass_on_speech_is_recorded(seed):
    """Two half passes everywhere (exact weights, activations as one half plane): 3-4e-5 on noise, at the edge on speech
    (7e-5 ... 1.1e-4 on 10 s chunks): not a mode bench.py may time as compliant. Bounded at 2e-4."""
    d = deviations("f16x2", seed, calibrate=False)
    print(f"f16x2 two passes everywhere, seed {seed}: " + ", ".join(f"{k} {v:.2e}" for k, v in d.items()))
    assert max(d.values()) <= 2e-4, d


# ----------------------------------------------------------------------------- calibrate_from_batchnorm(), numerically
def test_calibrate_from_batchnorm_matches_measured_calibration(tmp_path):
    """SequentialFromConfig(cfg, nnet3Path, gemm="f16x2") on a model whose BatchNorm statistics are its own (written through the
    Kaldi nnet3 binary format): calibrate_from_batchnorm() must give the statistics calibrate() measures on the same utterances,
    hence the same one-pass / residual-prefix route and the same x-vectors. Neither route is a tolerance-compliant mode: on this
    self-normalised network the form is at 1-2e-4 even on the calibration distribution, and further out on speech (which is why
    bench.py does not time it); the test pins the agreement of the two routes and the order of magnitude."""
    cal = np.concatenate([synth.make_wav(2, 160000, seed=777), synth.make_wav(2, 160000, seed=778, ragged=True)], 0)
    w = _self_consistent_weights(11, list(cal))
    synth.write_nnet3(str(tmp_path / "final.raw"), w)
    cfg = synth.extractor_cfg()
    seq = ktf.models.SequentialFromConfig(synth.model_config(), str(tmp_path / "final.raw"), "cmvn2xvec", gemm="f16x2")
    from_bn = seq.calibrate_from_batchnorm()
    assert len(from_bn) == 4                                    # tdnn2 .. tdnn5 read a BatchNorm'd ReLU plane
    a = ktf.models.XvectorExtractor.from_parts(cfg, seq, w["mean"], w["lda"])
    b = synth.build_extractor(ktf, cfg, w, gemm="f16x2")
    b.calibrate(torch.as_tensor(cal, device="cuda"))
    la = [l for l in a.xvec.layers if isinstance(l, ktf.layers.TDNN)]
    lb = [l for l in b.xvec.layers if isinstance(l, ktf.layers.TDNN)]
    for x, y in zip(la[1:5], lb[1:5]):
        ma, mb = a.xvec._xbar[id(x)], b.xvec._xbar[id(y)]
        assert np.abs(ma - mb).max() <= 2e-3 * max(1.0, np.abs(mb).max()), "BatchNorm mean != measured mean of the stored plane"
    noise = np.concatenate([synth.make_wav(1, 160000, seed=1234), synth.make_wav(1, 160000, seed=4242, ragged=True)], 0)
    whole, chunks = synth.speech_wavs()
    layers = synth.oracle_layers(w)
    for name, wav, bound in (("noise", noise, 5e-4), ("speech", chunks, 5e-3)):
        want = O.xvector_forward(wav, cfg, layers, w["mean"], w["lda"], dtype=np.float64)
        da = np.abs(a(torch.as_tensor(wav, device="cuda")).cpu().numpy() - want).max()
        db = np.abs(b(torch.as_tensor(wav, device="cuda")).cpu().numpy() - want).max()
        print(f"f16x2, BatchNorm statistics = own statistics: {name}: from BatchNorm {da:.2e}, measured {db:.2e}")
        assert da <= bound and db <= bound
        assert abs(da - db) <= 0.5 * max(da, db) + 2e-5, "the two calibration routes should agree"


def test_calibration_is_dropped_when_the_weights_change():
    w = synth.make_weights(seed=4321)
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm="f16x2", calibrate=True)
    assert mdl.xvec._xbar
    wav = torch.as_tensor(synth.make_wav(2, 32000, seed=5), device="cuda")
    run = mdl.compile(wav)
    run(wav)
    tdnn4 = mdl.xvec.get_layer("tdnn4.affine")
    tdnn4.set_weights([tdnn4.kaldi_matrix() * 1.5, tdnn4.bias])
    with pytest.raises(RuntimeError):
        run(wav)                                            # captured pointers / folded weights are stale
    mdl(wav)
    assert not mdl.xvec._xbar, "statistics measured on other weights must not survive set_weights"
