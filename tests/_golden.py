"""Loaders for tests/golden fixtures + the Kaldi-conf -> layer-kwargs mapping the
reference's testdata loaders use (testdata/feats/feats.py:42-77,95-130,147-167,184-212)."""

import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return _cache[name]


def _b(v):
    return v == "true"


def default_mfcc_cfg():
    """layers/dsp/mfcc_test.py:54-82."""
    return {
        "snip_edges": False,
        "framing": {"frame_length_ms": 25.0, "frame_shift_ms": 10.0, "sample_frequency": 16000.0},
        "mfcc": {
            "num_mfccs": 30, "num_mels": 30, "cepstral_lifter": 22, "use_energy": True,
            "sample_frequency": 16000.0, "high_freq_cutoff": 7600.0, "low_freq_cutoff": 20.0,
            "use_log_fbank": True, "use_power": True, "window_type": "povey", "dither": 0.0,
            "remove_dc_offset": True, "preemphasis_coefficient": 0.97, "raw_energy": True,
            "energy_floor": 0.0, "epsilon": float(np.finfo(np.float32).eps),
        },
    }


def mfcc_case(name):
    """-> (cfg, wav float32 (1,N) in int16 scale, want (1,T,C))."""
    z = load("feats_fbank_mfcc.npz")
    conf = json.loads(str(z["confs_json"]))[name]["mfcc"]
    cfg = default_mfcc_cfg()
    for k, v in conf.items():
        if k == "sample-frequency":
            cfg["framing"]["sample_frequency"] = float(v)
            cfg["mfcc"]["sample_frequency"] = float(v)
        elif k == "frame-length":
            cfg["framing"]["frame_length_ms"] = float(v)
        elif k == "frame-shift":
            cfg["framing"]["frame_shift_ms"] = float(v)
        elif k == "use-energy":
            cfg["mfcc"]["use_energy"] = _b(v)
        elif k == "raw-energy":
            cfg["mfcc"]["raw_energy"] = _b(v)
        elif k == "dither":
            cfg["mfcc"]["dither"] = float(v)
        elif k == "low-freq":
            cfg["mfcc"]["low_freq_cutoff"] = float(v)
        elif k == "high-freq":
            cfg["mfcc"]["high_freq_cutoff"] = float(v)
        elif k == "num-mel-bins":
            cfg["mfcc"]["num_mels"] = int(v)
        elif k == "num-ceps":
            cfg["mfcc"]["num_mfccs"] = int(v)
        elif k == "snip-edges":
            cfg["snip_edges"] = _b(v)
        else:
            raise ValueError(k)
    wav = z["wav_int16"].astype(np.float32).reshape(1, -1)
    return cfg, wav, z[f"mfcc_{name}"]


def default_fbank_cfg():
    """layers/dsp/filterbank_test.py defaultCfg."""
    return {
        "snip_edges": False,
        "framing": {"frame_length_ms": 25.0, "frame_shift_ms": 10.0, "sample_frequency": 16000.0},
        "windowing": {"window_type": "povey", "dither": 0.0, "remove_dc_offset": True,
                      "preemphasis_coefficient": 0.97, "raw_energy": True, "return_energy": False,
                      "energy_floor": 0.0, "epsilon": float(np.finfo(np.float32).eps)},
        "fbank": {"num_bins": 30, "sample_frequency": 16000.0, "high_freq_cutoff": -400.0, "low_freq_cutoff": 20.0,
                  "use_log_fbank": True, "use_power": True, "epsilon": float(np.finfo(np.float32).eps)},
    }


def fbank_case(name):
    z = load("feats_fbank_mfcc.npz")
    conf = json.loads(str(z["confs_json"]))[name]["fbank"]
    cfg = default_fbank_cfg()
    for k, v in conf.items():
        if k == "sample-frequency":
            cfg["framing"]["sample_frequency"] = float(v)
        elif k == "frame-length":
            cfg["framing"]["frame_length_ms"] = float(v)
        elif k == "frame-shift":
            cfg["framing"]["frame_shift_ms"] = float(v)
        elif k == "raw-energy":
            cfg["windowing"]["raw_energy"] = _b(v)
        elif k == "dither":
            cfg["windowing"]["dither"] = float(v)
        elif k == "low-freq":
            cfg["fbank"]["low_freq_cutoff"] = float(v)
        elif k == "high-freq":
            cfg["fbank"]["high_freq_cutoff"] = float(v)
        elif k == "num-mel-bins":
            cfg["fbank"]["num_bins"] = int(v)
        elif k == "use-log-fbank":
            cfg["fbank"]["use_log_fbank"] = _b(v)
        elif k == "use-power":
            cfg["fbank"]["use_power"] = _b(v)
        elif k == "snip-edges":
            cfg["snip_edges"] = _b(v)
        else:
            raise ValueError(k)
    wav = z["wav_int16"].astype(np.float32).reshape(1, -1)
    return cfg, wav, z[f"fbank_{name}"]


def fbank_case_names():
    z = load("feats_fbank_mfcc.npz")
    return sorted(k[len("fbank_"):] for k in z.files if k.startswith("fbank_"))


def mfcc_case_names():
    return [f"16000_{i:03d}" for i in range(1, 55)]


def vad_case(name):
    """-> (cfg, feats (1,T,D) = MFCC golden 16000_001, want mask (1,T,1))."""
    z = load("feats_vad.npz")
    conf = json.loads(str(z["confs_json"]))[name]
    cfg = {"energy_mean_scale": 0.5, "energy_threshold": 5.0, "frames_context": 0,
           "proportion_threshold": 0.6, "return_indexes": False, "energy_coeff": 0}
    m = {"vad-energy-threshold": ("energy_threshold", float), "vad-energy-mean-scale": ("energy_mean_scale", float),
         "vad-frames-context": ("frames_context", int), "vad-proportion-threshold": ("proportion_threshold", float)}
    for k, v in conf.items():
        key, f = m[k]
        cfg[key] = f(v)
    feats = load("feats_fbank_mfcc.npz")["mfcc_16000_001"]
    want = np.transpose(z[f"vad_{name}"], [0, 2, 1])
    return cfg, feats, want


def vad_case_names():
    return [f"16000_001_{i:03d}" for i in range(1, 47)]


def cmvn_case(name):
    z = load("feats_cmvn.npz")
    conf = json.loads(str(z["confs_json"]))[name]
    cfg = {"window": 600, "center": True, "norm_vars": False, "min_window": 100}
    m = {"cmn-window": ("window", int), "center": ("center", _b), "norm-vars": ("norm_vars", _b),
         "min-cmn-window": ("min_window", int)}
    for k, v in conf.items():
        key, f = m[k]
        cfg[key] = f(v)
    feats = load("feats_fbank_mfcc.npz")["mfcc_16000_001"]
    return cfg, feats, z[f"cmvn_{name}"]


def cmvn_case_names():
    return [f"16000_001_{i:03d}" for i in range(1, 9)]


STATS_CONFIGS = {   # layers/stats/stats_pooling_test.py:63-75 over defaultLayerCfg :34-46
    "stats_mean": {"include_std": False},
    "stats_mean_std": {},
    "stats_mean_std_windowed": {"right_context": 4},
    "stats_mean_std_only_left_context": {"left_context": -4, "right_context": 0},
    "stats_mean_std_both_left_right_context": {"left_context": -4, "right_context": 4},
    "stats_mean_std_asymmetrical_context": {"left_context": -4, "right_context": 2},
    "stats_mean_std_subsampling": {"input_period": 4, "output_period": 4},
    "stats_mean_std_windowed_subsampling": {"left_context": -4, "right_context": 4, "input_period": 4, "output_period": 4},
}


def stats_case(name):
    z = load("stats.npz")
    cfg = {"left_context": 0, "right_context": 16, "input_period": 1, "output_period": 1, "include_std": True,
           "padding": "SAME", "epsilon": 1e-10, "reduce_time_axis": False}
    cfg.update(STATS_CONFIGS[name])
    return cfg, z[f"{name}_in"], z[f"{name}_out"]


def narrow_layers():
    """tdnn_narrow model (layers/tdnn/tdnn_test.py:60-103) as oracle layer dicts + raw weights by name."""
    z = load("tdnn.npz")
    comps = json.loads(str(z["narrow_components_json"]))
    by = {}
    for c in comps:
        d = {}
        for k, v in c.items():
            d[k] = z[f"narrow_{c['name']}_{k}"] if v == "__array__" else v
        by[c["name"]] = d
    spec = [("tdnn1", [-2, -1, 0, 1, 2], True), ("tdnn2", [-2, 0, 2], True), ("tdnn3", [-3, 0, 3], True),
            ("tdnn4", [0], True), ("tdnn5", [0], True), ("output", [0], False)]
    layers = []
    for name, ctx, act in spec:
        a = by[f"{name}.affine"]
        layers.append({"kind": "tdnn", "name": f"{name}.affine", "W": a["params"], "b": a["bias"], "context": ctx})
        if act:
            layers.append({"kind": "relu", "name": f"{name}.relu"})
            bn = by[f"{name}.batchnorm"]
            layers.append({"kind": "bn", "name": f"{name}.batchnorm", "rms": bn["target-rms"],
                           "mean": bn["stats-mean"], "var": bn["stats-var"]})
    return layers, by, z["narrow_inputs"], z["narrow_outputs"]


def rmse(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
