#!/usr/bin/env python3
"""
Golden-vector generator (test infrastructure; runs only in the build container).

Reads the reference's Kaldi-generated golden data (text arks, *.conf files, binary
Kaldi objects, wavs) from /root/reference/kaldi_tflite/lib/testdata and imports the
reference's NumPy-only modules (kaldi_numpy, io readers, testdata constant tables)
to dump small fixtures under tests/golden/. Nothing from /root/reference is read at
test time; the GPU box never sees the reference.

Fixtures are DATA only: inputs, expected outputs, the Kaldi command-line options the
goldens were produced with, and binary Kaldi model files the reference's own tests
hold (nnet3 `final.raw`, `plda`, `mean.vec`, `transform.mat`).

Usage:  python tests/golden/make_golden.py
"""

import importlib.util
import json
import os
import shutil
import sys
import types

import numpy as np
from scipy.io import wavfile

REF = "/root/reference"
TD = os.path.join(REF, "kaldi_tflite/lib/testdata")
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference_numpy_parts():
    """Stub the parent packages so kaldi_tflite/__init__.py (which imports TF) never runs."""
    for n, p in [("kaldi_tflite", "kaldi_tflite"), ("kaldi_tflite.lib", "kaldi_tflite/lib")]:
        m = types.ModuleType(n)
        m.__path__ = [os.path.join(REF, p)]
        sys.modules[n] = m
    import kaldi_tflite.lib.io  # noqa: F401
    import kaldi_tflite.lib.kaldi_numpy  # noqa: F401
    return sys.modules["kaldi_tflite.lib.io"], sys.modules["kaldi_tflite.lib.kaldi_numpy"]


def load_py(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ark(path, dtype=np.float32):
    """Parse a Kaldi text ark (matrix or vector entries) -> dict[id] -> 2-D array."""
    ark, cur_id, cur = {}, None, []
    with open(path) as f:
        for line in f:
            toks = line.strip().split()
            if not toks:
                continue
            if "[" in toks and "]" in toks:
                if len(toks) > 3:
                    ark[toks[0]] = np.array([[float(t) for t in toks[2:-1]]], dtype=dtype)
                continue
            if "[" in toks:
                cur_id, cur = toks[0], []
                continue
            last = "]" in toks
            vals = [t for t in toks if t != "]"]
            if vals:
                cur.append([float(t) for t in vals])
            if last:
                ark[cur_id] = np.array(cur, dtype=dtype)
                cur_id, cur = None, []
    return ark


def parse_conf(path):
    cfg = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            k, v = line.split("=")
            cfg[k.lstrip("-")] = v
    return cfg


def read_wav_int16(path):
    sr, x = wavfile.read(path)
    assert x.dtype == np.int16 and x.ndim == 1, (x.dtype, x.shape)
    return sr, x


def main():
    ref_io, ref_np = import_reference_numpy_parts()
    os.chdir(REF)  # testdata python modules use cwd-relative paths

    # ---------------------------------------------------------------- feats
    sr, wav3s = read_wav_int16(os.path.join(TD, "librispeech_2_trimmed.wav"))
    assert sr == 16000
    out = {"wav_int16": wav3s}
    confs = {}
    for i in range(1, 55):
        name = f"16000_{i:03d}"
        d = os.path.join(TD, "feats/src/fbank_mfcc", name)
        confs[name] = {"mfcc": parse_conf(os.path.join(d, "mfcc.conf"))}
        out[f"mfcc_{name}"] = np.stack(list(load_ark(os.path.join(d, "mfcc.ark.txt")).values()), 0)
        if os.path.exists(os.path.join(d, "fbank.conf")):
            confs[name]["fbank"] = parse_conf(os.path.join(d, "fbank.conf"))
            out[f"fbank_{name}"] = np.stack(list(load_ark(os.path.join(d, "fbank.ark.txt")).values()), 0)
    out["confs_json"] = np.array(json.dumps(confs))
    np.savez_compressed(os.path.join(OUT, "feats_fbank_mfcc.npz"), **out)

    out, confs = {}, {}
    for i in range(1, 47):
        name = f"16000_001_{i:03d}"
        d = os.path.join(TD, "feats/src/vad", name)
        confs[name] = parse_conf(os.path.join(d, "vad.conf"))
        out[f"vad_{name}"] = np.stack(list(load_ark(os.path.join(d, "vad.ark.txt")).values()), 0)
    out["confs_json"] = np.array(json.dumps(confs))
    np.savez_compressed(os.path.join(OUT, "feats_vad.npz"), **out)

    out, confs = {}, {}
    for i in range(1, 9):
        name = f"16000_001_{i:03d}"
        d = os.path.join(TD, "feats/src/cmvn", name)
        confs[name] = parse_conf(os.path.join(d, "cmvn.conf"))
        out[f"cmvn_{name}"] = np.stack(list(load_ark(os.path.join(d, "cmvn.ark.txt")).values()), 0)
    out["confs_json"] = np.array(json.dumps(confs))
    np.savez_compressed(os.path.join(OUT, "feats_cmvn.npz"), **out)

    # ---------------------------------------------------------------- windowing (reference NumPy oracle)
    # windowing_test.py:85-120: frames = np.random.random((1,1000,256)) -> ProcessFrames(...)
    rng = np.random.RandomState(12345)
    frames = rng.random_sample((1, 40, 256))
    wcfgs = [
        {}, {"window_type": "hanning"}, {"window_type": "hamming"}, {"window_type": "rectangular"},
        {"window_type": "sine"}, {"window_type": "blackman"}, {"remove_dc_offset": False},
        {"preemphasis_coefficient": 0.0}, {"preemphasis_coefficient": 0.90}, {"raw_energy": False},
    ]
    out = {"frames": frames, "configs_json": np.array(json.dumps(wcfgs))}
    for i, o in enumerate(wcfgs):
        kw = dict(dither=0.0, remove_dc_offset=True, preemphasis_coefficient=0.97, window_type="povey", raw_energy=True)
        kw.update(o)
        win, en = ref_np.ProcessFrames(frames, **kw)
        out[f"windows_{i}"] = win
        out[f"energy_{i}"] = en
    # framing / padding helpers (framing_test.py:42-73)
    fcfgs = [(25.0, 10.0, 8000.0), (25.0, 10.0, 16000.0), (32.0, 16.0, 16000.0),
             (32.0, 32.0, 16000.0), (32.0, 64.0, 16000.0), (2000.0, 1000.0, 16000.0)]
    out["framing_configs"] = np.array(fcfgs)
    for i, (fl, fs, sf) in enumerate(fcfgs):
        N = int(10 * sf)
        m, k = int(sf * fl / 1000.0), int(sf * fs / 1000.0)
        x = np.arange(0, N)
        fr = ref_np.ExtractFrames(x, fl, fs, sf, True)
        xp = ref_np.PadWaveform(x, m, k)
        frp = ref_np.ExtractFrames(xp, fl, fs, sf, False)
        out[f"framing_{i}_snip_shape"] = np.array(fr.shape)
        out[f"framing_{i}_snip_first_col"] = np.ascontiguousarray(fr[:, 0])
        out[f"framing_{i}_pad_len"] = np.array(xp.shape[-1])
        out[f"framing_{i}_pad_head"] = xp[: 2 * m].copy()
        out[f"framing_{i}_pad_tail"] = xp[-2 * m:].copy()
        out[f"framing_{i}_nosnip_shape"] = np.array(frp.shape)
        out[f"framing_{i}_nosnip_first_col"] = np.ascontiguousarray(frp[:, 0])
        out[f"framing_{i}_nosnip_last_row"] = np.ascontiguousarray(frp[-1])
    # ApplyCMVN of the reference on a seeded input (cross-check of the restated helper)
    x = rng.standard_normal((1, 700, 5)).astype(np.float32) * 3 + 1
    out["cmvn_np_in"] = x
    for j, (w, nv, pad) in enumerate([(300, False, "SAME"), (300, True, "SAME"), (201, False, "VALID"), (900, True, "SAME")]):
        out[f"cmvn_np_out_{j}"] = ref_np.ApplyCMVN(x, center=True, norm_vars=nv, window=w, padding=pad)
    np.savez_compressed(os.path.join(OUT, "kaldi_numpy.npz"), **out)

    # ---------------------------------------------------------------- tdnn
    single = load_py(os.path.join(TD, "tdnn/tdnn_single_layer.py"), "ref_tdnn_single").RefTdnnSingleLayer
    narrow = load_py(os.path.join(TD, "tdnn/tdnn_narrow.py"), "ref_tdnn_narrow").RefTdnnNarrow
    W, b = single.weights()
    out = {
        "single_cfg_json": np.array(json.dumps(single.cfg)),
        "single_inputs": single.inputs, "single_outputs": single.outputs,
        "single_W": np.array(W), "single_b": np.array(b),
        "narrow_inputs": narrow.inputs, "narrow_outputs": narrow.outputs,
        "narrow_config_lines_json": np.array(json.dumps(narrow.config)),
    }
    comps = []
    for c in narrow.components:
        meta = {}
        for k, v in c.items():
            if isinstance(v, np.ndarray):
                out[f"narrow_{c['name']}_{k}"] = v
                meta[k] = "__array__"
            elif isinstance(v, (np.floating, np.integer)):
                meta[k] = v.item()
            else:
                meta[k] = v
        comps.append(meta)
    out["narrow_components_json"] = np.array(json.dumps(comps))
    # full-model goldens that need the (absent) pretrained final.raw; kept so tests can run when weights are supplied
    out["mfcc_chunk_30_16khz"] = np.stack(list(load_ark(os.path.join(TD, "mfcc_chunk_30_16khz.ark.txt")).values()), 0)
    out["tdnn6_affine_0008"] = np.stack(list(load_ark(os.path.join(TD, "tdnn/src/0008_sitw_v2_1a_tdnn6.affine/output.ark.txt")).values()), 0)
    out["tdnn6_affine_0006_feat"] = np.stack(list(load_ark(os.path.join(TD, "tdnn/src/0006_callhome_diarization_v2_1a_tdnn6.affine/feat.ark.txt")).values()), 0)
    out["tdnn6_affine_0006"] = np.stack(list(load_ark(os.path.join(TD, "tdnn/src/0006_callhome_diarization_v2_1a_tdnn6.affine/output.ark.txt")).values()), 0)
    np.savez_compressed(os.path.join(OUT, "tdnn.npz"), **out)
    shutil.copyfile(os.path.join(TD, "tdnn/src/tdnn_single_layer/final.raw"), os.path.join(OUT, "tdnn_single_layer.final.raw"))
    shutil.copyfile(os.path.join(TD, "tdnn/src/tdnn_narrow/final.raw"), os.path.join(OUT, "tdnn_narrow.final.raw"))

    # ---------------------------------------------------------------- stats pooling
    out = {}
    for name in ["stats_mean", "stats_mean_std", "stats_mean_std_windowed", "stats_mean_std_only_left_context",
                 "stats_mean_std_both_left_right_context", "stats_mean_std_asymmetrical_context",
                 "stats_mean_std_subsampling", "stats_mean_std_windowed_subsampling"]:
        d = os.path.join(TD, "stats/src", name)
        out[f"{name}_in"] = np.stack(list(load_ark(os.path.join(d, "feat.ark.txt")).values()), 0)
        out[f"{name}_out"] = np.stack(list(load_ark(os.path.join(d, "output.ark.txt")).values()), 0)
    np.savez_compressed(os.path.join(OUT, "stats.npz"), **out)

    # ---------------------------------------------------------------- plda / xvectors / lda
    xv = load_py(os.path.join(TD, "xvectors/xvectors.py"), "ref_xvectors").RefXVectors
    sc = load_py(os.path.join(TD, "plda/plda_scores.py"), "ref_plda_scores").RefPldaScores
    pm = load_py(os.path.join(TD, "plda/plda_model.py"), "ref_plda_model").RefPldaModel
    out = {
        "xvectors": np.stack(list(xv.ark.values()), 0),
        "xvector_mean": np.asarray(xv.mean),
        "plda_input": xv.pldaInput(),                       # (29,1,512) mean-sub + whitening + length norm
        "plda_transformed": xv.pldaTransformed(True),       # (29,512,1)
        "plda_scores": sc.scores(True),                     # (29,29)
        "plda_model_mean": np.asarray(pm.mean), "plda_model_psi": np.asarray(pm.psi),
        "plda_model_transform_row0": np.asarray(pm.transformMat)[0].copy(),
        "plda_model_transform_diag": np.ascontiguousarray(np.diag(np.asarray(pm.transformMat))),
    }
    np.savez_compressed(os.path.join(OUT, "plda.npz"), **out)
    shutil.copyfile(os.path.join(TD, "plda/plda"), os.path.join(OUT, "plda.bin"))
    for f in ["mean.vec", "mean.vec.txt", "transform.mat"]:
        shutil.copyfile(os.path.join(TD, "plda/xvectors_train_combined_200k", f), os.path.join(OUT, f"xvectors_train_combined_200k.{f}"))
    # first rows of the text matrix (array_reader_test compares text vs binary parse)
    with open(os.path.join(TD, "plda/xvectors_train_combined_200k/transform.mat.txt")) as f:
        lines = f.readlines()
    with open(os.path.join(OUT, "xvectors_train_combined_200k.transform.mat.head4.txt"), "w") as f:
        f.writelines(lines[:5])
        f.write("]\n") if "]" not in lines[4] else None

    # ---------------------------------------------------------------- e2e (needs pretrained weights to check; kept for when supplied)
    sr, wav22 = read_wav_int16(os.path.join(TD, "librispeech_2.wav"))
    d = os.path.join(TD, "models/src/0008_sitw_v2_1a")
    out = {
        "wav_int16": wav22, "sample_rate": np.array(sr),
        "xvector": np.stack(list(load_ark(os.path.join(d, "xvector.ark.txt")).values()), 0),
        "xvector_unnorm": np.stack(list(load_ark(os.path.join(d, "xvector.unnorm.ark.txt")).values()), 0),
        "confs_json": np.array(json.dumps({k: parse_conf(os.path.join(d, f"{k}.conf")) for k in ["mfcc", "vad", "cmvn"]})),
    }
    np.savez_compressed(os.path.join(OUT, "e2e_0008.npz"), **out)

    # ---------------------------------------------------------------- reader goldens: parsed values via the reference reader
    r = ref_io.KaldiNnet3Reader(os.path.join(TD, "tdnn/src/tdnn_narrow/final.raw"), True)
    out = {"config_json": np.array(json.dumps(r.config)), "names_json": np.array(json.dumps([[c["name"], c["type"]] for c in r.components]))}
    for c in r.components:
        for k, v in c.items():
            if isinstance(v, np.ndarray):
                out[f"{c['name']}|{k}"] = v
            elif isinstance(v, (np.floating, np.integer, float, int, bool, np.bool_)) and k not in ("name", "type"):
                out[f"{c['name']}|{k}"] = np.array(v)
    np.savez_compressed(os.path.join(OUT, "nnet3_narrow_parsed.npz"), **out)
    p = ref_io.KaldiPldaReader(os.path.join(TD, "plda/plda"), True)
    np.savez_compressed(os.path.join(OUT, "plda_parsed_head.npz"), mean=p.mean, psi=p.psi,
                        transform_row0=p.transformMat[0].copy(), transform_shape=np.array(p.transformMat.shape))

    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote fixtures to {OUT}: {tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
