"""Parity on the REAL pretrained Kaldi models — the reference's own end-to-end tests
(models/kaldi/sequential_test.py:70-117: tdnn6.affine of the 0008 / 0006 networks on a fixed MFCC chunk,
1 - cosine <= 1.25e-3; models/kaldi/xvector_extractor_test.py:70-96: wav -> x-vector of librispeech_2.wav, 1 - cosine
<= 0.075 against Kaldi's own embedding). The weights are not part of the reference repository (it downloads them from
kaldi-asr.org) and there is no network here, so like the reference's tests these SKIP unless the files are present:

    KTF_KALDI_MODELS=/path/to/kaldi_models        # holds 0008_sitw_v2_1a/exp/xvector_nnet_1a/final.raw (the extracted
                                                  # tarball) and optionally 0006_callhome_diarization_v2_1a/...

The Kaldi goldens themselves (tests/golden/tdnn.npz: mfcc_chunk_30_16khz, tdnn6_affine_*; tests/golden/e2e_0008.npz) are
committed. The CPU test runs the NumPy oracle on the real weights; the GPU tests run the HIP path and also compare it with
the oracle on the same weights.
"""

import json
import os

import numpy as np
import pytest

import _golden as G
import synth
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O

ROOT = os.environ.get("KTF_KALDI_MODELS", "")
MODELS = {  # name -> (feature dim, embedding dim, golden input key, golden output key)
    "0008_sitw_v2_1a": (30, 512, "mfcc_chunk_30_16khz", "tdnn6_affine_0008"),
    "0006_callhome_diarization_v2_1a": (23, 128, "tdnn6_affine_0006_feat", "tdnn6_affine_0006"),
}


def _raw(model):
    return os.path.join(ROOT, model, "exp", "xvector_nnet_1a", "final.raw")


def _need(model):
    if not ROOT or not os.path.exists(_raw(model)):
        pytest.skip(f"model weights not found (set KTF_KALDI_MODELS; wanted {_raw(model) if ROOT else '<unset>'})")


def cos_err(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return 1.0 - float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def _oracle_layers_from_raw(path):
    """oracle layer dicts (synth.oracle_layers' format) straight from a final.raw."""
    r = ktf.io.KaldiNnet3Reader(path, True)
    w = {}
    for name, _, _ in synth.TOPOLOGY:
        w[f"{name}.affine"] = tuple(r.getWeights(f"{name}.affine"))
        w[f"{name}.batchnorm"] = tuple(r.getWeights(f"{name}.batchnorm"))
    w["tdnn6.affine"] = tuple(r.getWeights("tdnn6.affine"))
    return synth.oracle_layers(w)


def test_harness_skips_cleanly_without_weights_and_goldens_are_committed():
    z = G.load("tdnn.npz")
    for model, (feat, out, kin, kout) in MODELS.items():
        assert z[kin].shape[-1] == feat and z[kout].shape == (1, 1, out)
    e = G.load("e2e_0008.npz")
    assert e["xvector"].shape == (1, 1, 128) and e["xvector_unnorm"].shape == (1, 1, 512) and e["wav_int16"].dtype == np.int16
    assert cos_err([1.0, 0.0], [1.0, 0.0]) == 0.0


@pytest.mark.parametrize("model", list(MODELS))
def test_oracle_tdnn6_affine_on_real_weights(model):
    _need(model)
    feat, out, kin, kout = MODELS[model]
    z = G.load("tdnn.npz")
    got = O.sequential_forward(_oracle_layers_from_raw(_raw(model)), z[kin], dtype=np.float32)
    assert got.shape == z[kout].shape
    assert cos_err(z[kout], got) <= 1.25e-3                          # sequential_test.py:30


# the modes that claim the tolerance on any input
REAL_MODES = ["f32", "bf16x3", "f16mx"]


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", REAL_MODES)
@pytest.mark.parametrize("model", list(MODELS))
def test_gpu_tdnn6_affine_on_real_weights(model, gemm):
    _need(model)
    import torch
    feat, out, kin, kout = MODELS[model]
    z = G.load("tdnn.npz")
    mdl = ktf.models.SequentialFromConfig(synth.model_config(False, feat, out), _raw(model), "cmvn2xvec", gemm=gemm)
    mdl.min_tiles = {}
    got = mdl(z[kin]).cpu().numpy()
    assert got.shape == z[kout].shape
    assert cos_err(z[kout], got) <= 1.25e-3
    want = O.sequential_forward(_oracle_layers_from_raw(_raw(model)), z[kin], dtype=np.float64)
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", REAL_MODES)
def test_gpu_wav_to_xvector_on_real_weights(gemm, tmp_path):
    model = "0008_sitw_v2_1a"
    _need(model)
    import torch
    import yaml
    e = G.load("e2e_0008.npz")
    mdir = os.path.join(ROOT, model, "exp", "xvector_nnet_1a")
    # the extractor YAML of the reference (data/tflite_models/0008_sitw_v2_1a.yml) with the Kaldi recipe's front-end options
    # (dither off so the comparison is deterministic) and the paths of the supplied tarball
    ecfg = synth.extractor_cfg(dither=0.0)
    (tmp_path / "kaldi.yml").write_text(yaml.safe_dump({"name": model, "model_config": synth.model_config()}))
    mean_p = os.path.join(mdir, "xvectors_train_combined_200k", "mean.vec")
    lda_p = os.path.join(mdir, "xvectors_train_combined_200k", "transform.mat")
    if not (os.path.exists(mean_p) and os.path.exists(lda_p)):       # the two small files are also committed fixtures
        mean_p = os.path.join(G.GOLDEN, "xvectors_train_combined_200k.mean.vec.txt")
        lda_p = os.path.join(G.GOLDEN, "xvectors_train_combined_200k.transform.mat")
    ecfg["xvec"] = {"model_config_path": str(tmp_path / "kaldi.yml"), "model_path": _raw(model),
                    "global_mean_path": mean_p, "lda_matrix_path": lda_p}
    (tmp_path / "extractor.yml").write_text(yaml.safe_dump({"name": model, "extractor": ecfg}))
    mdl = ktf.models.XvectorExtractorFromConfig(str(tmp_path / "extractor.yml"), gemm=gemm)
    mdl.xvec.min_tiles = {}
    wav = e["wav_int16"].astype(np.float32)[None]
    got = mdl(torch.as_tensor(wav, device="cuda")).cpu().numpy()
    assert got.shape == (128,)
    assert cos_err(e["xvector"], got) <= 0.075                       # xvector_extractor_test.py:30
    want = O.xvector_forward(wav, ecfg, _oracle_layers_from_raw(_raw(model)), mdl.xvecGlobalMean,
                             np.concatenate([mdl.ldaMat.T, mdl.ldaOffset.T], 1), dtype=np.float64)
    assert np.abs(got - want[0]).max() <= 1e-4       # north_star bound, on the real weights
