"""CPU-only tests: the C-ABI library loads and exports every symbol the header declares, argument validation
works without a GPU, and the host-side logic (Kaldi readers, kaldi_numpy helpers, config builders, weight import,
layer constructors) behaves like the reference. No kernel is launched here."""

import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import yaml

import _golden as G
import synth
import kaldi_tflite_amd as ktf
from kaldi_tflite_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- C-ABI
def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "ktf_hip.h")).read()
    declared = set(re.findall(r"\b(ktf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ktf_version() == 117


def test_abi_argument_validation_without_gpu():
    lib = L.load()
    cfg = L.FrontendCfg(frame_size=400, frame_shift=160, nfft=500, num_mels=30, num_ceps=30)
    tab = L.FrontendTables()
    rc = lib.ktf_frontend_f32(None, 1, 16000, 0, C.byref(cfg), C.byref(tab), 3, None, None, 0, None)
    assert rc == -1 and "null" in L.last_error()
    buf = (C.c_float * 4)()
    rc = lib.ktf_frontend_f32(buf, 1, 16000, 0, C.byref(cfg), C.byref(tab), 3, buf, None, 0, None)
    assert rc == -1 and "power of two" in L.last_error()
    with pytest.raises(ValueError):
        L.check(rc, "x")
    d = L.TdnnDesc(units=8, din=3, din_pad=32, nctx=2, subsampling=1)
    d.ctx[0], d.ctx[1] = 1, -1
    rc = lib.ktf_tdnn(buf, 1, 8, 32, None, C.byref(d), buf, None, None, None, None, buf, 8, None, None)
    assert rc == -1 and "ascending" in L.last_error()
    v = L.VadCfg(energy_threshold=5, energy_mean_scale=0.5, proportion_threshold=0.6, frames_context=2, energy_coeff=40)
    assert lib.ktf_vad_mask_f32(buf, 1, 10, 30, C.byref(v), buf, None) == -1
    # split-bf16 plane entry points: mode / dtype / plane checks happen before any launch
    d = L.TdnnDesc(units=512, din=64, din_pad=64, nctx=1, subsampling=1, gemm=L.GEMM_BF16, x_dtype=L.KTF_BF16,
                   w_dtype=L.KTF_BF16, y_dtype=L.KTF_BF16)
    rc = lib.ktf_tdnn_split(buf, buf, 1, 8, 64, None, C.byref(d), buf, buf, None, None, None, buf, buf, 512, None, None)
    assert rc == -1 and "KTF_GEMM_BF16X3" in L.last_error()
    d.gemm = L.GEMM_BF16X3
    rc = lib.ktf_tdnn_split(buf, None, 1, 8, 64, None, C.byref(d), buf, buf, None, None, None, buf, buf, 512, None, None)
    assert rc == -1 and "lo plane" in L.last_error()
    d.units = 64                                                    # the plane route runs on the 256x256 kernels only
    rc = lib.ktf_tdnn_split(buf, buf, 1, 8, 64, None, C.byref(d), buf, buf, None, None, None, buf, buf, 64, None, None)
    assert rc == -1 and "units > 128" in L.last_error()
    d.units, d.valid = 512, 1
    rc = lib.ktf_tdnn_split_stats(buf, buf, 1, 8, 64, None, C.byref(d), buf, buf, None, None, None, buf, None)
    assert rc == -1 and "SAME" in L.last_error()
    assert lib.ktf_split_bf16(buf, 4, 0, 8, buf, buf, 8, None) == -1 and "bad sizes" in L.last_error()
    assert lib.ktf_split_bf16(buf, 4, 4, 8, buf, None, 8, None) == -1 and "null" in L.last_error()
    assert lib.ktf_split_bf16(buf, 0, 4, 8, buf, buf, 8, None) == 0   # nothing to do


def test_host_side_size_helpers():
    lib = L.load()
    assert lib.ktf_num_frames(160000, 400, 160) == 998          # 10 s @ 16 kHz (Framing does not pad)
    assert lib.ktf_num_frames(48000, 400, 160) == 298
    assert lib.ktf_num_frames(399, 400, 160) == 0
    # pad_mode 1 = the reference's kaldi_numpy PadWaveform followed by Framing (frame_extraction.py:54-89)
    for n, m, k in [(160000, 400, 160), (48077, 400, 160), (8013, 200, 80), (1000, 400, 160), (23, 400, 160)]:
        fr = ktf.layers.Framing(1000.0 * m / 16000, 1000.0 * k / 16000, 16000, snip_edges=False)
        try:
            want = ktf.layers.Framing(1000.0 * m / 16000, 1000.0 * k / 16000, 16000).numFrames(
                ktf.kaldi_numpy.PadWaveform(np.zeros(n, np.float32), m, k).shape[-1])
        except ValueError:
            want = None
        got = lib.ktf_num_frames_padded(n, m, k, 1)
        if got >= 0:
            assert got == want == fr.numFrames(n), (n, m, k)
        else:
            with pytest.raises(ValueError):
                fr.numFrames(n)
    assert lib.ktf_num_frames_padded(160000, 400, 160, 0) == 998
    d = L.TdnnDesc(units=8, din=3, din_pad=32, nctx=3, subsampling=1, valid=1)
    d.ctx[0], d.ctx[1], d.ctx[2] = -2, 0, 2
    assert lib.ktf_tdnn_out_len(10, C.byref(d)) == 6
    d.valid = 0
    d.subsampling = 3
    assert lib.ktf_tdnn_out_len(10, C.byref(d)) == 4


def test_gemm_mode_selection_host_logic():
    import torch
    assert L.ktf_dtype(torch.float32) == L.KTF_F32 and L.ktf_dtype(torch.bfloat16) == L.KTF_BF16
    assert L.act_torch_dtype(L.GEMM_BF16) == torch.bfloat16 and L.act_torch_dtype(L.GEMM_BF16X3) == torch.float32
    big = ktf.layers.TDNN(512, context=[-2, 0, 2], gemm="f16mx")
    assert big.effective_gemm(L.GEMM_F16MX, relu=True) == L.GEMM_F16MX and big.effective_gemm(L.GEMM_BF16) == L.GEMM_BF16
    # the block-scaled mode runs on the MX kernels only: narrow layers and sigmoid / tanh layers are evaluated in fp32
    assert ktf.layers.TDNN(64, context=[0], gemm="f16mx").effective_gemm(L.GEMM_F16MX) == L.GEMM_F32
    assert ktf.layers.TDNN(512, context=[0], activation="tanh", gemm="f16mx").effective_gemm(L.GEMM_F16MX) == L.GEMM_F32
    assert ktf.layers.TDNN(512, context=[0], activation="relu", gemm="f16mx").effective_gemm(L.GEMM_F16MX, relu=False) == L.GEMM_F16MX
    big.build((None, None, 512))
    d = big.desc(L.GEMM_BF16, torch.bfloat16, torch.bfloat16, act="relu")
    assert (d.x_dtype, d.w_dtype, d.y_dtype, d.gemm) == (L.KTF_BF16, L.KTF_BF16, L.KTF_BF16, L.GEMM_BF16)
    for gone in ("f16", "f16x2", "float16"):             # round 2's half-precision modes left the library in round 5
        with pytest.raises(ValueError):
            ktf.layers.TDNN(8, context=[0], gemm=gone)
    with pytest.raises(ValueError):
        ktf.layers.TDNN(8, context=[0], gemm="fp8")
    # batches of only a few 256-row tiles leave the 256-row kernels in every reduced mode (Sequential.batch_gemm: the crossovers of
    # tools/small_batch_crossover.py) -- for the bf16-pair small tiles (`small_tile_pairs`, the default) or the exact fp32 ones
    S = ktf.models.Sequential
    assert S([], gemm="bf16").batch_gemm(2, 998) == L.GEMM_F32 and S([], gemm="bf16").batch_gemm(3, 998) == L.GEMM_BF16
    assert S([], gemm="bf16").batch_gemm(14, 200) == L.GEMM_F32 and S([], gemm="bf16").batch_gemm(15, 200) == L.GEMM_BF16   # (rows / 256, rounded up)
    assert S([], gemm="bf16x3").batch_gemm(7, 998) == L.GEMM_F32 and S([], gemm="bf16x3").batch_gemm(8, 998) == L.GEMM_BF16X3
    assert S([], gemm="f16mx").batch_gemm(4, 998) == L.GEMM_F32 and S([], gemm="f16mx").batch_gemm(5, 998) == L.GEMM_F16MX
    assert S([], gemm="f32").batch_gemm(1024, 998) == L.GEMM_F32
    assert S([], gemm="f16mx")._batch_route(1, 998) == (L.GEMM_F32, True)          # small tiles, pairs behind the first layer
    assert S([], gemm="f16mx")._batch_route(1024, 998) == (L.GEMM_F16MX, False)
    assert S([], gemm="f16mx")._batch_route(1024, 300) == (L.GEMM_BF16X3, False)    # short utterances: the tighter mode, its own kernels
    assert S([], gemm="f16mx")._batch_route(2, 300) == (L.GEMM_F32, True)
    assert S([], gemm="f32")._batch_route(1, 998) == (L.GEMM_F32, False)            # the exact mode never leaves the fp32 kernels
    m = S([], gemm="bf16")
    m.min_tiles = {}
    assert m.batch_gemm(1, 10) == L.GEMM_BF16
    # f16mx: which kernel takes a batch of B x 998 frames (tools/mid_batch.py measured the loader-wave kernel faster exactly there)
    mm = S([], gemm="f16mx")
    assert [b for b in (5, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256, 1024) if mm._mx_use_loader(b, 998)] == [5, 6, 8, 12, 16, 24, 48]
    mm.mx_loader = True
    assert mm._mx_use_loader(1024, 998)
    mm.mx_loader = False
    assert not mm._mx_use_loader(6, 998)
    m = S([], gemm="f16mx")
    m.small_tile_pairs = False
    assert m._batch_route(1, 998) == (L.GEMM_F32, False)
    # KTF_BF16P: operands of KTF_GEMM_BF16X4, in float32 tensors
    assert L.ktf_dtype(L.PAIR) == L.KTF_BF16P and L.act_torch_dtype(L.GEMM_BF16X4) == torch.float32
    from kaldi_tflite_amd import ops
    v = torch.tensor([0.0, 1.0, -3.14159265, 1e-3, 65504.0, 1e-30])
    p = ops.pair_encode(v)
    bits = p.view(torch.int32)
    assert p.dtype == torch.float32 and int(bits[0]) == 0 and (int(bits[1]) & 0xFFFF) == 0x3F80 and (int(bits[1]) >> 16) == 0
    assert torch.all((ops.pair_decode(p) - v).abs() <= v.abs() * 2.0 ** -16)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ktf.KtfBackendError):
        ktf.layers.Framing()(np.arange(16000, dtype=np.float32))


# ----------------------------------------------------------------------------- readers (io/kaldi/*_test.py of the reference)
def test_nnet3_reader_matches_reference_parse():
    z = G.load("nnet3_narrow_parsed.npz")
    r = ktf.io.KaldiNnet3Reader(os.path.join(G.GOLDEN, "tdnn_narrow.final.raw"), True)
    assert r.config == json.loads(str(z["config_json"]))
    assert [[c["name"], c["type"]] for c in r.components] == json.loads(str(z["names_json"]))
    n = 0
    for c in r.components:
        for k, v in c.items():
            key = f"{c['name']}|{k}"
            if key in z.files:
                assert np.array_equal(np.asarray(v), z[key]), key
                n += 1
    assert n >= 30
    w = r.getWeights("tdnn1.affine")
    assert len(w) == 2 and w[0].shape == (5, 15) and w[1].shape == (5,)
    bn = r.getWeights("tdnn1.batchnorm")
    assert len(bn) == 3 and np.ndim(bn[0]) == 0
    with pytest.raises(KeyError):
        r.getWeights("nonexistent")
    # same values as the python-literal copy the reference keeps (nnet3_reader_test.py)
    _, by, _, _ = G.narrow_layers()
    assert np.abs(w[0] - by["tdnn1.affine"]["params"]).max() <= 1e-7


def test_nnet3_reader_single_layer_file():
    r = ktf.io.KaldiNnet3Reader(os.path.join(G.GOLDEN, "tdnn_single_layer.final.raw"), True)
    z = G.load("tdnn.npz")
    assert np.array_equal(r.components[0]["params"], z["single_W"])
    assert np.array_equal(r.components[0]["bias"], z["single_b"])


def test_text_mode_object_reader_not_supported():
    with pytest.raises(NotImplementedError):
        ktf.io.KaldiObjReader(os.path.join(G.GOLDEN, "xvectors_train_combined_200k.mean.vec.txt"), False)


def test_plda_reader():
    z = G.load("plda_parsed_head.npz")
    p = ktf.io.KaldiPldaReader(os.path.join(G.GOLDEN, "plda.bin"), True)
    assert tuple(z["transform_shape"]) == p.transformMat.shape == (512, 512)
    assert np.array_equal(p.mean, z["mean"]) and np.array_equal(p.psi, z["psi"])
    assert np.array_equal(p.transformMat[0], z["transform_row0"])
    lit = G.load("plda.npz")
    assert np.abs(p.mean - lit["plda_model_mean"]).max() <= 1e-9        # plda_reader_test.py tolerance
    assert np.abs(np.diag(p.transformMat) - lit["plda_model_transform_diag"]).max() <= 1e-9


def test_read_kaldi_array_binary_and_text(tmp_path):
    vb = ktf.io.ReadKaldiArray(os.path.join(G.GOLDEN, "xvectors_train_combined_200k.mean.vec"), binary=True)
    vt = ktf.io.ReadKaldiArray(os.path.join(G.GOLDEN, "xvectors_train_combined_200k.mean.vec.txt"), binary=False)
    assert vb.shape == vt.shape == (512,)
    assert G.rmse(vb, vt) <= 5e-8                                        # array_reader_test.py tolerance
    mb = ktf.io.ReadKaldiArray(os.path.join(G.GOLDEN, "xvectors_train_combined_200k.transform.mat"), binary=True)
    assert mb.shape == (128, 513)
    head = ktf.io.ReadKaldiArray(os.path.join(G.GOLDEN, "xvectors_train_combined_200k.transform.mat.head4.txt"), binary=False)
    flat = head.reshape(-1) if head.ndim == 2 else np.concatenate([np.asarray(r) for r in head])
    assert G.rmse(flat, mb.reshape(-1)[: flat.size]) <= 5e-7
    # round trip through our own writers (matrix with several rows, text)
    m = np.arange(12, dtype=np.float32).reshape(3, 4)
    p = tmp_path / "m.txt"
    p.write_text(" [\n  0 1 2 3\n  4 5 6 7\n  8 9 10 11 ]\n")
    assert np.array_equal(ktf.io.ReadKaldiArray(str(p), binary=False), m)
    synth.write_bin_mat(str(tmp_path / "m.bin"), m)
    assert np.array_equal(ktf.io.ReadKaldiArray(str(tmp_path / "m.bin"), binary=True), m)
    (tmp_path / "bad.bin").write_bytes(b"\0BXX 1234")
    with pytest.raises(ValueError):
        ktf.io.ReadKaldiArray(str(tmp_path / "bad.bin"), binary=True)
    (tmp_path / "open.txt").write_text(" [\n 1 2 3\n")
    with pytest.raises(ValueError):
        ktf.io.ReadKaldiArray(str(tmp_path / "open.txt"), binary=False)


# ----------------------------------------------------------------------------- kaldi_numpy
def test_kaldi_numpy_matches_reference_outputs():
    z = G.load("kaldi_numpy.npz")
    frames = z["frames"]
    for i, o in enumerate(json.loads(str(z["configs_json"]))):
        kw = dict(dither=0.0, remove_dc_offset=True, preemphasis_coefficient=0.97, window_type="povey", raw_energy=True)
        kw.update(o)
        w, e = ktf.kaldi_numpy.ProcessFrames(frames, **kw)
        assert np.allclose(w, z[f"windows_{i}"], rtol=0, atol=1e-12)
        assert np.allclose(e, z[f"energy_{i}"], rtol=0, atol=1e-12)
    for i, (fl, fs, sf) in enumerate(z["framing_configs"]):
        N = int(10 * sf)
        m, k = int(sf * fl / 1000.0), int(sf * fs / 1000.0)
        x = np.arange(0, N)
        fr = ktf.kaldi_numpy.ExtractFrames(x, fl, fs, sf, True)
        assert fr.shape == tuple(z[f"framing_{i}_snip_shape"]) and np.array_equal(fr[:, 0], z[f"framing_{i}_snip_first_col"])
        xp = ktf.kaldi_numpy.PadWaveform(x, m, k)
        assert xp.shape[-1] == int(z[f"framing_{i}_pad_len"])
        assert np.array_equal(xp[: 2 * m], z[f"framing_{i}_pad_head"]) and np.array_equal(xp[-2 * m:], z[f"framing_{i}_pad_tail"])
        frp = ktf.kaldi_numpy.ExtractFrames(xp, fl, fs, sf, False)
        assert frp.shape == tuple(z[f"framing_{i}_nosnip_shape"]) and np.array_equal(frp[-1], z[f"framing_{i}_nosnip_last_row"])
    x = z["cmvn_np_in"]
    for j, (w, nv, pad) in enumerate([(300, False, "SAME"), (300, True, "SAME"), (201, False, "VALID"), (900, True, "SAME")]):
        got = ktf.kaldi_numpy.ApplyCMVN(x, center=True, norm_vars=nv, window=w, padding=pad)
        assert np.allclose(got, z[f"cmvn_np_out_{j}"], rtol=0, atol=1e-6)
    with pytest.raises(NotImplementedError):
        ktf.kaldi_numpy.ApplyCMVN(x)
    # window helper (frame_extraction.py:141-187): the windowed frames above pin "povey"; names and errors here
    n = np.arange(400)
    assert np.allclose(ktf.kaldi_numpy.GetWindowFunction("povey", 400), (0.5 - 0.5 * np.cos(2 * np.pi * n / 399)) ** 0.85)
    assert np.array_equal(ktf.kaldi_numpy.GetWindowFunction("hamming", 25), np.hamming(25))
    for bad in [("povey", 0), ("triangle", 25)]:
        with pytest.raises(ValueError):
            ktf.kaldi_numpy.GetWindowFunction(*bad)
    x2 = x.reshape(-1, x.shape[-1])
    xp = np.pad(x2, [[1, 0], [0, 0]])
    s = ktf.kaldi_numpy.getWindowedSums(xp, 11, "VALID")
    assert s.shape[0] == x2.shape[0] - 10 and np.allclose(s[3], x2[3:14].sum(0))
    ss = ktf.kaldi_numpy.getWindowedSums(xp, 11, "SAME")
    assert ss.shape == x2.shape and np.array_equal(ss[0], ss[5]) and np.allclose(ss[5], s[0]) and np.array_equal(ss[-1], ss[-6])


# ----------------------------------------------------------------------------- layer constructors / config errors
def test_layer_constructor_errors_match_reference():
    Ls = ktf.layers
    for kw in [{"frame_length_ms": 0}, {"frame_shift_ms": -1}, {"sample_frequency": 0}, {"frame_length_ms": 0.01}]:
        with pytest.raises(ValueError):
            Ls.Framing(**kw)
    with pytest.raises(ValueError):
        Ls.Framing().build((None, None))
    with pytest.raises(ValueError):
        Ls.Framing().build((1, 100))
    with pytest.raises(ValueError):
        Ls.Windowing(preemphasis_coefficient=1.5)
    with pytest.raises(ValueError):
        Ls.Windowing(window_type="kaiser")
    for kw in [{"num_bins": 2}, {"sample_frequency": 0}, {"low_freq_cutoff": 9000}, {"low_freq_cutoff": 500, "high_freq_cutoff": 400}]:
        with pytest.raises(ValueError):
            Ls.FilterBank(**kw)
    with pytest.raises(NotImplementedError):
        Ls.DCT(10, dct_type=3)
    with pytest.raises(NotImplementedError):
        Ls.DCT(10, norm="none")
    with pytest.raises(ValueError):
        Ls.DCT(0)
    with pytest.raises(ValueError):
        Ls.DCT(40).build((1, 10, 30))
    with pytest.raises(ValueError):
        Ls.MFCC(num_mfccs=31, num_mels=30)
    for kw in [{"energy_mean_scale": -1}, {"frames_context": -1}, {"proportion_threshold": 0}, {"proportion_threshold": 1}]:
        with pytest.raises(ValueError):
            Ls.VAD(**kw)
    with pytest.raises(NotImplementedError):
        Ls.CMVN(center=False)
    with pytest.raises(ValueError):
        Ls.CMVN(window=0)
    with pytest.raises(ValueError):
        Ls.CMVN(padding="causal")
    with pytest.raises(ValueError):
        Ls.TDNN(8, subsampling_factor=0)
    with pytest.raises(ValueError):
        Ls.TDNN(8, padding="causal")
    with pytest.raises(ValueError):
        Ls.TDNN(8, context="x")
    for kw in [{"left_context": 1, "right_context": 0}, {"left_context": 0, "right_context": -1},
               {"left_context": 0, "right_context": 1, "input_period": 0},
               {"left_context": 0, "right_context": 1, "input_period": 2, "output_period": 3}]:
        with pytest.raises(ValueError):
            Ls.StatsPooling(**kw)
    with pytest.raises(AssertionError):
        Ls.PLDA(4, np.zeros(3), np.eye(4), np.ones(4))
    with pytest.raises(AssertionError):
        Ls.PLDA(4, np.zeros(4), np.zeros((4, 3)), np.ones(4))


def test_layer_shapes_configs_and_weights():
    Ls = ktf.layers
    f = Ls.Framing(25.0, 10.0, 16000.0)
    assert (f.frameSize, f.frameShift, f.numFrames(160000)) == (400, 160, 998)
    assert f.compute_output_shape([1, 48000]) == [1, 298, 400]
    assert f.get_config()["frame_length"] == 25.0
    t = Ls.TDNN(32, context=[1, -3, 0, -1], activation="relu", name="t")
    assert t.context == [-3, -1, 0, 1] and t.kernelWidth == 4
    t2 = Ls.TDNN.from_config({"units": 32, "context": [-3, -1, 0, 1], "subsampling_factor": 1, "padding": "SAME",
                              "activation": "relu", "use_bias": True})          # tdnn_test.py:40
    z = G.load("tdnn.npz")
    t2.build((1, 8, 30))
    t2.set_weights([z["single_W"], z["single_b"]])
    assert t2.kernel.shape == (1, 4, 30, 32)
    assert np.array_equal(t2.kaldi_matrix(), z["single_W"])
    assert np.array_equal(t2.kernel, Ls.reshapeKaldiTdnnWeights(z["single_W"], 32, 4))
    assert t2.kernel[0, 2, 7, 5] == z["single_W"][5, 2 * 30 + 7]
    with pytest.raises(ValueError):
        t2.set_weights([z["single_W"]])
    with pytest.raises(ValueError):
        t2.set_weights([z["single_W"], z["single_b"]], fmt="onnx")
    t2.set_weights(t2.get_weights(), fmt="tensorflow")
    assert t2.compute_output_shape((1, 8, 30)) == (1, 8, 32)
    tv = Ls.TDNN(4, context=[-2, 0, 2], padding="VALID", subsampling_factor=2)
    assert tv.compute_output_shape((1, 11, 3)) == (1, 4, 4)
    bn = Ls.BatchNorm(name="bn")
    bn.set_weights([np.float32(2.0), np.array([1.0, 2.0], np.float32), np.array([4.0, 9.0], np.float32)])
    g, m, v = bn.get_weights()
    assert np.array_equal(g, [2.0, 2.0])
    s, h = bn.affine()
    assert np.allclose(s, 2.0 / np.sqrt(np.array([4.0, 9.0]) + 1e-3)) and np.allclose(h, -m * s)
    with pytest.raises(ValueError):
        bn.set_weights([1.0, m])
    c = Ls.CMVN(window=300, padding="VALID")
    assert c.compute_output_shape([1, 998, 30]) == [1, 998 - 299, 30]
    sp = Ls.StatsPooling(0, 10000, reduce_time_axis=True)
    assert sp.compute_output_shape((2, 998, 1500)) == (2, 1, 3000)
    spv = Ls.StatsPooling(-4, 4, padding="VALID")
    assert spv.compute_output_shape((1, 16, 3)) == (1, 9, 6)
    m = Ls.MFCC(num_mfccs=30, num_mels=30)
    m.build((1, 998, 400))
    assert m._cfg.nfft == 512 and m.filterbank.melBank.shape == (257, 30) and m.dct.dct.shape == (30, 30)
    assert np.all(m.filterbank.melBank[256] == 0)


def test_mel_dct_tables_equal_oracle():
    from oracle import ktf_oracle as O
    from kaldi_tflite_amd import ops
    for nb, hi in [(23, -400.0), (30, 7600.0), (40, 0.0)]:
        hi_abs = hi if hi > 0 else hi + 8000.0
        nfft, bank = ops.mel_bank_dense(400, nb, 16000.0, 20.0, hi_abs)
        nfft2, bank2 = O.mel_bank(400, nb, 16000.0, hi, 20.0)
        assert nfft == nfft2 == 512 and np.array_equal(bank, bank2)
    assert np.array_equal(ops.dct_matrix(30, 30), O.dct_matrix(30, 30))
    assert np.array_equal(ops.lifter_coeffs(30, 22), O.lifter_coeffs(30, 22))
    for w in ["povey", "hamming", "hanning", "rectangular", "sine", "blackman"]:
        assert np.array_equal(ops.window_function(w, 400), O.window_function(w, 400))


# ----------------------------------------------------------------------------- builders (sequential_test.py / xvector_extractor.py)
def test_sequential_from_config_errors():
    S = ktf.models.SequentialFromConfig
    with pytest.raises(ValueError):
        S({"layers": []})
    with pytest.raises(ValueError):
        S({"layers": [{"name": "tdnn1", "type": "affine", "cfg": {"units": 4}}]})
    with pytest.raises(ValueError):
        S({"layers": [{"name": "input", "type": "input", "shape": [None, None, 3]}, {"name": "x", "type": "conv9"}]})
    with pytest.raises(KeyError):
        S({"layers": [{"name": "input", "type": "input", "shape": [None, None, 3]}, {"name": "x"}]})


def test_sequential_from_config_with_nnet3_file(tmp_path):
    w = synth.make_weights(seed=7, narrow=True)
    path = str(tmp_path / "final.raw")
    synth.write_nnet3(path, w)
    r = ktf.io.KaldiNnet3Reader(path, True)
    assert len(r.components) == 18 and r.components[-1]["name"] == "tdnn6.affine"
    mdl = ktf.models.SequentialFromConfig(synth.model_config(narrow=True), path, "cmvn2xvec")
    names = [l.name for l in mdl.layers]
    assert names[:3] == ["tdnn1.affine", "tdnn1.relu", "tdnn1.batchnorm"] and names[-2:] == ["stats", "tdnn6.affine"]
    assert len(names) == 17
    for name in ["tdnn1", "tdnn3", "tdnn5"]:
        W, b = w[f"{name}.affine"]
        l = mdl.get_layer(f"{name}.affine")
        assert np.array_equal(l.kaldi_matrix(), W) and np.array_equal(l.bias, b)
        rms, mean, var = w[f"{name}.batchnorm"]
        bn = mdl.get_layer(f"{name}.batchnorm")
        assert np.array_equal(bn.moving_mean, mean) and np.array_equal(bn.moving_variance, var) and np.all(bn.gamma == rms)
    assert np.array_equal(mdl.get_layer("tdnn6.affine").kaldi_matrix(), w["tdnn6.affine"][0])
    plan = mdl._plan()
    assert [s[0] for s in plan] == ["tdnn"] * 5 + ["stats", "tdnn"] and all(s[2] and s[3] is not None for s in plan[:5])


def test_xvector_extractor_from_config_builds(tmp_path):
    w = synth.make_weights(seed=9, narrow=True)
    mdir = tmp_path / "0008_sitw_v2_1a" / "exp" / "xvector_nnet_1a"
    (mdir / "xvectors_train_combined_200k").mkdir(parents=True)
    synth.write_nnet3(str(mdir / "final.raw"), w)
    synth.write_text_vec(str(mdir / "xvectors_train_combined_200k" / "mean.vec"), w["mean"])
    synth.write_bin_mat(str(mdir / "xvectors_train_combined_200k" / "transform.mat"), w["lda"])
    kcfg = {"name": "0008_sitw_v2_1a", "sample_rate": 16000,
            "download": {"link": "https://kaldi-asr.org/models/8/0008_sitw_v2_1a.tar.gz", "hash": "0" * 64},
            "model_config": synth.model_config(narrow=True)}
    (tmp_path / "kaldi.yml").write_text(yaml.safe_dump(kcfg))
    ecfg = synth.extractor_cfg()
    ecfg["xvec"] = {"model_config_path": str(tmp_path / "kaldi.yml"), "model_path": str(mdir / "final.raw"),
                    "global_mean_path": str(mdir / "xvectors_train_combined_200k" / "mean.vec"),
                    "lda_matrix_path": str(mdir / "xvectors_train_combined_200k" / "transform.mat")}
    (tmp_path / "extractor.yml").write_text(yaml.safe_dump({"name": "0008_sitw_v2_1a", "extractor": ecfg}))
    mdl = ktf.models.XvectorExtractorFromConfig(str(tmp_path / "extractor.yml"))
    assert mdl.ldaMat.shape == (512, 128) and mdl.ldaOffset.shape == (1, 128) and mdl.xvecGlobalMean.shape == (512,)
    assert np.array_equal(mdl.ldaMat, w["lda"][:, :-1].T) and np.allclose(mdl.xvecGlobalMean, w["mean"])
    assert np.array_equal(mdl.xvec.get_layer("tdnn2.affine").kaldi_matrix(), w["tdnn2.affine"][0])
    assert (mdl.framing.frameSize, mdl.mfcc.numMfccs, mdl.vad.framesContext, mdl.cmvn.N) == (400, 30, 2, 300)
    os.remove(str(mdir / "final.raw"))
    with pytest.raises(FileNotFoundError):
        ktf.models.XvectorExtractorFromConfig(str(tmp_path / "extractor.yml"))


def test_reference_yaml_topology_is_parseable():
    # the two nnet3 topologies the reference ships (0008 16 kHz/30-dim, 0006 8 kHz/23-dim/128-out), restated as dicts
    for feat, out in [(30, 512), (23, 128)]:
        cfg = synth.model_config()
        cfg["layers"][0]["shape"] = [None, None, feat]
        cfg["layers"][-1]["cfg"]["units"] = out
        mdl = ktf.models.SequentialFromConfig(cfg, None, "cmvn2xvec")
        assert mdl.get_layer("tdnn1.affine").kernel.shape == (1, 5, feat, 512)
        assert mdl.get_layer("tdnn6.affine").kernel.shape == (1, 1, 3000, out)


def test_shipped_yaml_configs_build_the_reference_topologies():
    # kaldi_tflite_amd/data/**/*.yml (tools/write_model_configs.py): the YAML files a user of the reference passes to the builders
    data = os.path.join(os.path.dirname(ktf.__file__), "data")
    for name, feat, out, right in [("0008_sitw_v2_1a", 30, 512, 10000), ("0006_callhome_diarization_v2_1a", 23, 128, 400)]:
        with open(os.path.join(data, "kaldi_models", "configs", f"{name}.yml")) as f:
            cfg = yaml.safe_load(f)
        assert cfg["name"] == name and cfg["download"]["link"].endswith(f"{name}.tar.gz") and len(cfg["download"]["hash"]) == 64
        mdl = ktf.models.SequentialFromConfig(cfg["model_config"], None, "cmvn2xvec")
        assert [l.name for l in mdl.layers][-2:] == ["stats", "tdnn6.affine"] and len(mdl.layers) == 17
        assert mdl.get_layer("tdnn1.affine").kernel.shape == (1, 5, feat, 512)
        assert mdl.get_layer("tdnn6.affine").kernel.shape == (1, 1, 3000, out)
        assert mdl.get_layer("stats").rightContext == right and mdl.get_layer("tdnn3.affine").context == [-3, 0, 3]
    with open(os.path.join(data, "tflite_models", "0008_sitw_v2_1a.yml")) as f:
        ext = yaml.safe_load(f)["extractor"]
    assert ext == dict(synth.extractor_cfg(dither=1.0), xvec=ext["xvec"])
    assert ext["xvec"]["model_config_path"] == "data/kaldi_models/configs/0008_sitw_v2_1a.yml"


def test_split_plane_weight_layouts_follow_the_header():
    """KTF_TDNN_K_INTERLEAVED / KTF_TDNN_W_TILED as include/ktf_hip.h defines them, checked on the host packing
    (TDNN.device_weights needs a GPU for the upload only: the packing is re-derived here from the kernel matrix)."""
    rng = np.random.default_rng(0)
    U, K, D = 300, 3, 40                                   # -> units_pad 512, Din_pad 64, ktot 192
    t = ktf.layers.TDNN(U, context=[-2, 0, 2])
    t.build((1, 8, D))
    W = rng.standard_normal((U, K * D)).astype(np.float32)
    t.set_weights([W, np.zeros(U, np.float32)])
    Up, Dp = 512, 64
    ref = np.zeros((Up, K, Dp))
    ref[:U, :, :D] = W.reshape(U, K, D)
    # K-interleaved: column ((d / 32) * nctx + k) * 32 + d % 32 holds W[u, k * Din_pad + d]
    inter = np.ascontiguousarray(ref.reshape(Up, K, Dp // 32, 32).transpose(0, 2, 1, 3)).reshape(Up, K * Dp)
    for u, k, d in [(0, 0, 0), (7, 2, 39), (299, 1, 33), (100, 2, 5)]:
        assert inter[u, ((d // 32) * K + k) * 32 + d % 32] == W[u, k * D + d]
    # tiled: block (nt, ks) = 16 KiB of halves; row r, 16-byte position q holds columns 32 ks + 8 (q ^ ((4 - (r >> 2)) & 3)) .. + 7
    nt, nks = Up // 256, K * Dp // 32
    r = np.arange(256)
    src = np.arange(4)[None, :] ^ ((4 - ((r >> 2) & 3)) & 3)[:, None]
    tiled = np.ascontiguousarray(inter.reshape(nt, 256, nks, 4, 8)[:, r[:, None], :, src, :].transpose(2, 3, 0, 1, 4)).reshape(-1)
    for n_, ks, row, q in [(0, 0, 0, 0), (1, 5, 43, 2), (0, 3, 255, 3), (1, 2, 17, 1), (0, 1, 6, 0)]:
        sw = (4 - ((row >> 2) & 3)) & 3
        want = inter[n_ * 256 + row, ks * 32 + 8 * (q ^ sw): ks * 32 + 8 * (q ^ sw) + 8]
        off = ((n_ * nks + ks) * 16384 + 64 * row + 16 * q) // 2             # halves
        assert np.array_equal(tiled[off: off + 8], want)
    # the same code path the model uses (source inspection keeps the two in step: the packing lines are these)
    import inspect
    body = inspect.getsource(ktf.layers.TDNN.device_weights)
    assert "transpose(0, 2, 1, 3)" in body and "(4 - ((r >> 2) & 3)) & 3" in body and "transpose(2, 3, 0, 1, 4)" in body


def test_library_kernel_families():
    """The TDNN GEMM kernel instantiations shipped in libktf_hip.so are exactly the ones the dispatcher can reach (the map itself is
    pinned on the GPU by tests/test_gpu_dispatch.py): no probe / ablation / superseded generations in the product."""
    import collections
    import re
    import subprocess
    from kaldi_tflite_amd import _lib
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    kernels = set(re.findall(r"__device_stub__(tdnn_\w+(?:<[^(]*>)?)\(", out))
    fam = collections.Counter(k.split("<")[0] for k in kernels)
    assert set(fam) == {"tdnn_f32_kernel", "tdnn_f32s_kernel", "tdnn_f32t_kernel", "tdnn_f32_rowvec_kernel", "tdnn_bf16_kernel",
                        "tdnn_bf16g_kernel", "tdnn_bf16r_kernel", "tdnn_bf16r16_kernel", "tdnn_bf16h_kernel", "tdnn_x3r_kernel",
                        "tdnn_x3s_kernel", "tdnn_x4s_kernel", "tdnn_mx_kernel", "tdnn_mxl_kernel",
                        "tdnn_out_lens_kernel"}, fam          # (the last one: ktf_tdnn_out_lens, lengths only)
    assert fam["tdnn_x3r_kernel"] == 8                  # 4 activations x {store, pooled}: fp32 activations only
    assert fam["tdnn_x3s_kernel"] == 8 + 8 + 4          # split-bf16 (4 activations x {rows, pooled}, plain and row-group-skipping; flat rows: {ReLU, none} x {rows, pooled})
    assert fam["tdnn_x4s_kernel"] == 8                  # bf16-pair small tiles: 64 x 32 / 64 / 96 and the K-step-32 form, x {rows, pooled}
    assert fam["tdnn_mx_kernel"] == 12 + 8              # {ReLU, none} x {planes, fp32, pooled} x {K-steps fill the super-steps, padded} + flat row tiles: {ReLU, none} x {planes, pooled} x {...}
    assert fam["tdnn_mxl_kernel"] == 6                  # the same on the loader-wave kernel (KTF_TDNN_MX_LOADER)
    assert not any("probe" in k for k in kernels)
    assert "getenv" not in out


def test_abi_argument_validation_from_c(tmp_path):
    """tests/abi_validation.c: a plain C caller hands every TDNN / MX / tail / helper entry point arguments it must refuse; each
    call has to come back KTF_EINVAL with a message, before any HIP call (runs without a GPU). tools/asan_abi.sh runs the same
    driver against an AddressSanitizer + UBSan build of the library (built in ~30 s on 8 cores and cached under $TMPDIR, keyed by a
    checksum of the sources; KTF_SKIP_SANITIZERS=1 skips it)."""
    import os
    import subprocess
    from kaldi_tflite_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_validation")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "abi_validation.c"), "-o", exe,
                    "-L" + libdir, "-l:libktf_hip.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "all rejected as KTF_EINVAL" in out.stdout, out.stdout + out.stderr
    if os.environ.get("KTF_SKIP_SANITIZERS") != "1" and os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        san = subprocess.run(["bash", os.path.join(root, "tools", "asan_abi.sh")], capture_output=True, text=True, timeout=900)
        assert san.returncode == 0 and "all rejected as KTF_EINVAL" in san.stdout and "ERROR: AddressSanitizer" not in san.stderr, san.stdout + san.stderr


def test_sequential_with_a_valid_padded_windowed_pooling_builds_for_any_number_of_frames():
    """Found by tools/fuzz_models.py: StatsPooling.compute_output_shape compared the unknown time axis of a dynamically shaped model with
    the window width (TypeError at SequentialFromConfig); the time axis stays unknown and the layer behind sees the doubled width."""
    import kaldi_tflite_amd as ktf
    cfg = {"type": "sequential", "layers": [
        {"name": "input", "type": "input", "shape": [None, None, 30]},
        {"name": "t0", "type": "affine", "cfg": {"units": 96, "context": [2], "subsampling_factor": 3}},
        {"name": "w", "type": "stats", "cfg": {"left_context": -2, "right_context": 4, "input_period": 2, "output_period": 2, "padding": "VALID"}},
        {"name": "t1", "type": "affine", "cfg": {"units": 16, "context": [0]}}]}
    m = ktf.models.SequentialFromConfig(cfg, None, "m")
    assert m.get_layer("t1.affine").inputDim == 192
    assert m.get_layer("w").compute_output_shape((None, None, 96)) == (None, None, 192)
    assert m.get_layer("w").compute_output_shape((2, 40, 96)) == (2, m.get_layer("w").numOutputSteps(40), 192)
