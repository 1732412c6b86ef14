"""GPU parity on NON-STATIONARY audio (run with `-m gpu` on an MI355X): every GEMM arithmetic mode that claims the north_star
tolerance (<= 1e-4 max-abs x-vector deviation) is checked on the reference's own end-to-end input -- 22.5 s of read speech
(testdata/librispeech_2.wav, models/kaldi/xvector_extractor_test.py:70-96; committed as tests/golden/e2e_0008.npz:wav_int16) --
whole and as 10 s chunks, plus amplitude-modulated coloured noise, over several weight seeds, against the fp64 oracle. The
stationary noise of the throughput workload cannot show errors that depend on the input distribution: a constant bias per
utterance survives the statistics pooling, zero-mean rounding noise does not (round 2's calibrated one-pass half mode passed on
noise and was 4-7e-4 on speech: it left the library in round 5)."""

import numpy as np
import pytest
import torch

import synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O

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


def deviations(gemm, seed, prepare=None):
    w = synth.make_weights(seed=seed)
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm=gemm)
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


def test_a_captured_graph_refuses_to_replay_after_set_weights():
    w = synth.make_weights(seed=4321)
    mdl = synth.build_extractor(ktf, synth.extractor_cfg(), w, gemm="f16mx")
    wav = torch.as_tensor(synth.make_wav(2, 32000, seed=5), device="cuda")
    run = mdl.compile(wav)
    run(wav)
    tdnn4 = mdl.xvec.get_layer("tdnn4.affine")
    tdnn4.set_weights([tdnn4.kaldi_matrix() * 1.5, tdnn4.bias])
    with pytest.raises(RuntimeError):
        run(wav)                                            # captured pointers / folded weights are stale
    mdl(wav)
