"""Test helper (uses the fp64 oracle: test infrastructure only): synthetic 0008 weights whose BatchNorm statistics are the network's
own activation statistics -- the closest offline stand-in for a trained Kaldi model's <StatsMean> / <StatsVar> (the pretrained
final.raw is not shipped with the reference; models/kaldi/download.py:28-100 fetches it)."""

import numpy as np

import synth
from oracle import ktf_oracle as O


def self_consistent_weights(seed, cal_wavs):
    """Synthetic 0008 weights whose BatchNorm statistics ARE the network's own activation statistics on `cal_wavs` (what a
    trained Kaldi model's <StatsMean> / <StatsVar> are): one fp64 oracle pass, layer by layer -- the ReLU outputs' mean / variance
    become that layer's BatchNorm statistics before the pass continues through it."""
    w = synth.make_weights(seed=seed)
    cfg = synth.extractor_cfg()
    feats = []
    fcfg = {k: v for k, v in cfg["framing"].items() if k != "dynamic_input_shape"}
    for wav in cal_wavs:
        fr = O.framing(wav[None].astype(np.float64), **fcfg)
        m = O.mfcc(fr, **cfg["mfcc"], dtype=np.float64)
        vcfg = dict(cfg["vad"]); vcfg["return_indexes"] = True
        idx = O.vad(m, **vcfg, dtype=np.float64)
        feats.append(O.cmvn(m[idx[:, 0], idx[:, 1]][None], **cfg["cmvn"], dtype=np.float64))
    xs = feats
    for name, ctx, _ in synth.TOPOLOGY:
        W, b = w[f"{name}.affine"]
        xs = [O.relu(O.tdnn(x, W, b, list(ctx), dtype=np.float64)) for x in xs]
        allv = np.concatenate([x[0] for x in xs], 0)
        mean, var = allv.mean(0), allv.var(0)
        w[f"{name}.batchnorm"] = (np.float32(1.0), mean.astype(np.float32), var.astype(np.float32))
        xs = [O.batchnorm(x, 1.0, mean.astype(np.float32), var.astype(np.float32), 1e-3, dtype=np.float64) for x in xs]
    return w
