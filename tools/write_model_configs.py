#!/usr/bin/env python3
"""Writes the model / extractor YAML files the package ships (kaldi_tflite_amd/data/...): the two x-vector topologies the
reference supports (Kaldi recipes 0008_sitw_v2_1a and 0006_callhome_diarization_v2_1a: layer widths, contexts, pooling,
download location and checksum of the public kaldi-asr.org tarballs) and the 0008 extractor (the front-end options of the
recipe's mfcc.conf / vad.conf / cmvn.conf). The schema is the one SequentialFromConfig / XvectorExtractorFromConfig read; the
relative paths inside follow the reference's layout (run from a directory that holds data/kaldi_models/<name>/...)."""
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "kaldi-tflite_amd", "kaldi_tflite_amd", "data")

MODELS = {
    "0008_sitw_v2_1a": dict(desc="x-vector DNN trained on augmented VoxCeleb 1 + 2 (Kaldi recipe sitw/v2)", rate=16000, feat=30,
                            embed=512, pool_right=10000, url="https://kaldi-asr.org/models/8/0008_sitw_v2_1a.tar.gz",
                            sha256="0e056069866c53421751fb58dabe1f1dc3dec05ca196aa67b970d9ec261bda60"),
    "0006_callhome_diarization_v2_1a": dict(desc="x-vector DNN trained on augmented Switchboard + NIST SREs (Kaldi recipe callhome_diarization/v2)",
                                            rate=8000, feat=23, embed=128, pool_right=400,
                                            url="https://kaldi-asr.org/models/6/0006_callhome_diarization_v2_1a.tar.gz",
                                            sha256="4ebcd87c4239de073c19ad5a0d782b60d10344d4d2d5a58cc39fb91ccbbbb461"),
}
FRAME_LAYERS = [("tdnn1", 512, [-2, -1, 0, 1, 2]), ("tdnn2", 512, [-2, 0, 2]), ("tdnn3", 512, [-3, 0, 3]), ("tdnn4", 512, [0]), ("tdnn5", 1500, [0])]


def model_yaml(name, m):
    layers = [{"name": "input", "type": "input", "shape": [None, None, m["feat"]]}]
    layers += [{"name": n, "type": ["affine", "relu", "batchnorm"], "cfg": {"units": u, "context": c}} for n, u, c in FRAME_LAYERS]
    layers.append({"name": "stats", "type": "stats_pooling",
                   "cfg": {"left_context": 0, "right_context": m["pool_right"], "include_std": True, "reduce_time_axis": True}})
    layers.append({"name": "tdnn6", "type": "affine", "cfg": {"units": m["embed"], "context": [0]}})
    return {"name": name, "description": m["desc"], "sample_rate": m["rate"], "download": {"link": m["url"], "hash": m["sha256"]},
            "model_config": {"type": "sequential", "layers": layers}}


def extractor_yaml(name):
    base = f"data/kaldi_models/{name}/exp/xvector_nnet_1a"
    return {"name": name, "description": MODELS[name]["desc"],
            "extractor": {
                "framing": {"frame_length_ms": 25, "frame_shift_ms": 10, "sample_frequency": 16000, "dynamic_input_shape": True},
                "mfcc": {"num_mfccs": 30, "num_mels": 30, "sample_frequency": 16000.0, "high_freq_cutoff": 7600.0,
                         "low_freq_cutoff": 20.0, "dither": 1.0},
                "vad": {"energy_mean_scale": 0.5, "energy_threshold": 5.5, "frames_context": 2, "proportion_threshold": 0.12,
                        "return_indexes": True, "energy_coeff": 0},
                "cmvn": {"center": True, "norm_vars": False, "window": 300},
                "xvec": {"model_config_path": f"data/kaldi_models/configs/{name}.yml", "model_path": f"{base}/final.raw",
                         "global_mean_path": f"{base}/xvectors_train_combined_200k/mean.vec",
                         "lda_matrix_path": f"{base}/xvectors_train_combined_200k/transform.mat"}},
            "scorer": {"plda": {"model_path": f"{base}/xvectors_train_combined_200k/plda", "dim": 128, "normalize_length": True,
                                "simple_length_norm": False, "return_transformed": False}}}


def main():
    os.makedirs(os.path.join(OUT, "kaldi_models", "configs"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "tflite_models"), exist_ok=True)
    for name, m in MODELS.items():
        with open(os.path.join(OUT, "kaldi_models", "configs", f"{name}.yml"), "w") as f:
            yaml.safe_dump(model_yaml(name, m), f, sort_keys=False, default_flow_style=None)
    with open(os.path.join(OUT, "tflite_models", "0008_sitw_v2_1a.yml"), "w") as f:
        yaml.safe_dump(extractor_yaml("0008_sitw_v2_1a"), f, sort_keys=False, default_flow_style=None)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
