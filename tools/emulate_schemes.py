#!/usr/bin/env python3
"""CPU-only fp64 emulation of reduced-precision TDNN arithmetic schemes on noise AND speech (measurement tool; uses the oracle
as the checker, never shipped). A scheme says, per frame-level layer, how the stored input plane is rounded and how the
(BatchNorm-folded) weights are represented; everything else is float64. Prints the max-abs x-vector deviation from the exact
network per (input, weight seed).

    python tools/emulate_schemes.py [scheme ...]        (no arguments: the list at the bottom)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, synth
from oracle import ktf_oracle as O

cfg = synth.extractor_cfg()
EPS = 1e-3


def features(wav):
    """(N,) samples -> CMVN'd voiced features (T', 30) float64 (oracle front-end)."""
    fcfg = {k: v for k, v in cfg["framing"].items() if k != "dynamic_input_shape"}
    fr = O.framing(wav[None].astype(np.float64), **fcfg)
    m = O.mfcc(fr, **cfg["mfcc"], dtype=np.float64)
    vcfg = dict(cfg["vad"]); vcfg["return_indexes"] = True
    idx = O.vad(m, **vcfg, dtype=np.float64)
    return O.cmvn(m[idx[:, 0], idx[:, 1]][None], **cfg["cmvn"], dtype=np.float64)[0]


def splice(x, ctx):
    T = x.shape[0]
    t = np.arange(T)
    return np.concatenate([x[np.clip(t + c, 0, T - 1)] for c in ctx], 1)


def h16(a):
    return a.astype(np.float16).astype(np.float64)


def bf(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def quant_block(a, mant_bits, exp_bits, block=32):
    """Block-scaled minifloat along the last axis (MX style): per `block` elements a power-of-two scale chosen so that the
    block maximum lands in the top binade; elements keep `mant_bits` stored mantissa bits and `exp_bits` exponent bits
    (values below the smallest normal become subnormal multiples). Round to nearest."""
    sh = a.shape
    pad = (-sh[-1]) % block
    if pad:
        a = np.concatenate([a, np.zeros(sh[:-1] + (pad,))], -1)
    v = a.reshape(-1, block)
    mx = np.abs(v).max(1, keepdims=True)
    e_top = np.floor(np.log2(np.where(mx > 0, mx, 1.0)))
    emin = e_top - (2 ** exp_bits - 2)                       # smallest normal exponent within the block's range
    e = np.floor(np.log2(np.where(v != 0, np.abs(v), 1.0)))
    e = np.maximum(e, emin)
    q = 2.0 ** (e - mant_bits)
    r = np.round(v / q) * q
    r = r.reshape(a.shape)
    return r[..., : sh[-1]] if pad else r


def mx(a, fmt, block=32, scale_from=None, scale_shift=0):
    """OCP MX block format along the last axis with saturation: fmt 'e2m1' (fp4: max 6), 'e2m3' (fp6: max 7.5), 'e3m2' (fp6: max 28),
    'e4m3' (fp8: max 448). Scale = 2^(floor(log2(blockmax)) - emax), bumped by one when the maximum would round past the largest
    element value. `scale_from`: take the block maxima from this array instead (times 2^scale_shift): a derived scale."""
    mb, emax, vmax = {"e2m1": (1, 2, 6.0), "e2m3": (3, 2, 7.5), "e3m2": (2, 4, 28.0), "e4m3": (3, 8, 448.0)}[fmt]
    sh = a.shape
    pad = (-sh[-1]) % block
    if pad:
        a = np.concatenate([a, np.zeros(sh[:-1] + (pad,))], -1)
        if scale_from is not None:
            scale_from = np.concatenate([scale_from, np.zeros(sh[:-1] + (pad,))], -1)
    v = a.reshape(-1, block)
    src = v if scale_from is None else scale_from.reshape(-1, block) * 2.0 ** scale_shift
    mxv = np.abs(src).max(1, keepdims=True)
    e_top = np.floor(np.log2(np.where(mxv > 0, mxv, 1.0)))
    sc = 2.0 ** (e_top - emax)
    # largest value must not round beyond vmax
    half_ulp_top = 2.0 ** (emax - mb - 1)
    sc = np.where(mxv / sc >= vmax + half_ulp_top, sc * 2, sc)
    u = v / sc
    e = np.floor(np.log2(np.where(u != 0, np.abs(u), 1.0)))
    e = np.maximum(e, 0.0)                         # exponent of the smallest normal binade is 0 (value 1.0); below: subnormal step 2^-mb
    q = 2.0 ** (e - mb)
    r = np.clip(np.round(u / q) * q, -vmax, vmax) * sc
    r = r.reshape(a.shape)
    return r[..., : sh[-1]] if pad else r


PLANE_Q = {
    "exact": lambda a: a,
    "h": h16,
    "bf": bf,
    "h+h": lambda a: h16(a) + h16(a - h16(a)),
    "h+e4m3": lambda a: h16(a) + quant_block(a - h16(a), 3, 4),
    "h+e2m3": lambda a: h16(a) + quant_block(a - h16(a), 3, 2),
    "h+e3m2": lambda a: h16(a) + quant_block(a - h16(a), 2, 3),
    "h+e2m1": lambda a: h16(a) + quant_block(a - h16(a), 1, 2),
}


def run(feats, w, scheme, cal=None, collect=None):
    """scheme: list of 5 (plane_q, wmode) for tdnn1..tdnn5. wmode: 'exact' | 'h' (nearest half, no correction) |
    'h_cal' (nearest half + bias correction with the calibration mean cal[i]) | 'h_utt' (… with this utterance's own mean) |
    'h+e2m3' etc.: half main + block-scaled minifloat residual weights (along K), the residual product taken with the
    PLANE's minifloat image (x6 * wlo6) -- emulated as exact x times rounded residual plus the x rounding on that term."""
    x = feats
    pend = None                     # BatchNorm of the previous layer, folded forward
    for i, (name, ctx, _) in enumerate(synth.TOPOLOGY):
        W, b = (np.asarray(a, np.float64) for a in w[f"{name}.affine"])
        rms, mean, var = (np.asarray(a, np.float64) for a in w[f"{name}.batchnorm"])
        D = x.shape[1]
        if pend is not None:        # W' = W diag(s), b' = b + W h   (s, h per input feature, tiled over the contexts)
            s, hsh = pend
            st, ht = np.tile(s, len(ctx)), np.tile(hsh, len(ctx))
            b = b + W @ ht
            W = W * st[None, :]
        pq, wm = scheme[i]
        xs = PLANE_Q[pq](x)
        if collect is not None:
            collect.setdefault(i, []).append(xs)
        X = splice(xs, ctx)
        if wm == "exact":
            z = X @ W.T + b
        elif wm in ("h", "h_cal", "h_utt"):
            Wr = h16(W)
            bb = b
            if wm == "h_cal":
                bb = b - (Wr - W) @ np.tile(cal[i], len(ctx))
            elif wm == "h_utt":
                bb = b - (Wr - W) @ np.tile(xs.mean(0), len(ctx))
            z = X @ Wr.T + bb
        elif wm.startswith("h+"):
            mb, eb = {"e4m3": (3, 4), "e2m3": (3, 2), "e3m2": (2, 3), "e2m1": (1, 2)}[wm[2:]]
            Wh = h16(W)
            Wl = quant_block(W - Wh, mb, eb)
            Xq = quant_block(splice(h16(x), ctx), mb, eb)       # the plane's minifloat image multiplies the residual weights
            z = X @ Wh.T + Xq @ Wl.T + b
        elif wm.startswith("mx:"):
            # full scheme: X_h W_h + Q(X_lo) Q(W) + Q(X_h) Q(W_lo); wm = "mx:<fmt of the x_lo*W term>:<fmt of the x*W_lo term>[:derived]"
            parts = wm.split(":")
            f1, f2 = parts[1], parts[2]
            derived = len(parts) > 3 and parts[3] == "d"
            f1x, f1w = (f1.split("/") * 2)[:2]
            f2x, f2w = (f2.split("/") * 2)[:2]
            xh = h16(x)
            Xh = splice(xh, ctx)
            Wh = h16(W)
            z = Xh @ Wh.T + b
            if f1 != "none":
                xlo = x - xh if pq != "h" else np.zeros_like(x)
                xlo = (x - xh)
                if derived:
                    Xlo = mx(splice(xlo, ctx), f1x, scale_from=Xh, scale_shift=-11)
                else:
                    Xlo = mx(splice(xlo, ctx), f1x)
                z = z + Xlo @ mx(W, f1w).T
            if f2 != "none":
                if f2x == "trunc8":          # top byte of the half value: e5m2 by truncation
                    Xq = (Xh.astype(np.float16).view(np.uint16) & 0xFF00).view(np.float16).astype(np.float64)
                else:
                    Xq = mx(Xh, f2x)
                z = z + Xq @ mx(W - Wh, f2w).T
        else:
            raise ValueError(wm)
        x = np.maximum(z, 0.0)
        pend = (rms / np.sqrt(var + EPS), -mean * rms / np.sqrt(var + EPS))
    s, hsh = pend
    x = x * s + hsh                                   # last frame-level layer: BatchNorm in the epilogue (fp32 there)
    pooled = O.stats_pooling(x[None], left_context=0, right_context=10000, include_std=True, reduce_time_axis=True, dtype=np.float64)
    W6, b6 = w["tdnn6.affine"]
    hvec = pooled.reshape(1, -1) @ np.asarray(W6, np.float64).T + np.asarray(b6, np.float64)
    return O.xvector_post(hvec.reshape(1, 1, -1), w["mean"], w["lda"], dtype=np.float64).reshape(-1)


def inputs():
    z = np.load(os.path.join(ROOT, "tests", "golden", "e2e_0008.npz"))
    sp = z["wav_int16"].astype(np.float64)
    rng = np.random.default_rng(99)
    n = 160000
    white = rng.standard_normal(n)
    f = np.fft.rfftfreq(n, 1 / 16000.0)
    pink = np.fft.irfft(np.fft.rfft(white) / np.sqrt(np.maximum(f, 20.0)), n)
    pink *= 1000.0 / pink.std()
    am = pink * (0.55 + 0.45 * np.sin(2 * np.pi * 3.0 * np.arange(n) / 16000.0)) * (1 + 0.5 * np.sin(2 * np.pi * 0.31 * np.arange(n) / 16000.0))
    return {
        "noise": synth.make_wav(1, n, seed=1234)[0].astype(np.float64),
        "noise_ragged": synth.make_wav(1, n, seed=4242, ragged=True)[0].astype(np.float64),
        "speech22": sp,
        "speech10a": sp[:n],
        "speech10b": sp[n:2 * n],
        "pink_am": np.round(am),
    }


SCHEMES = {
    "two_pass": [("h", "exact")] * 5,
    "one_plane_P0": [("h", "exact")] + [("exact", "exact")] * 4,
    "one_plane_P1": [("exact", "exact"), ("h", "exact")] + [("exact", "exact")] * 3,
    "one_plane_P2": [("exact", "exact")] * 2 + [("h", "exact")] + [("exact", "exact")] * 2,
    "one_plane_P3": [("exact", "exact")] * 3 + [("h", "exact")] + [("exact", "exact")],
    "one_plane_P4": [("exact", "exact")] * 4 + [("h", "exact")],
    "t1_exact_in": [("h+h", "exact")] + [("h", "exact")] * 4,
    "tail_utt": [("h", "exact")] * 3 + [("h", "h_utt")] * 2,
    "tail_cal": [("h", "exact")] * 3 + [("h", "h_cal")] * 2,
    "x_e2m3": [("h+h", "exact")] + [("h+e2m3", "exact")] * 4,
    "x_e4m3": [("h+h", "exact")] + [("h+e4m3", "exact")] * 4,
    "x_e2m1": [("h+h", "exact")] + [("h+e2m1", "exact")] * 4,
    "xw_e2m3": [("h+h", "exact")] + [("h+e2m3", "h+e2m3")] * 4,
    "w_e2m3": [("h+h", "exact")] + [("h", "h+e2m3")] * 4,
    "mx44": [("exact", "mx:e2m1:e2m1")] * 5,
    "mx44d": [("exact", "mx:e2m1:e2m1:d")] * 5,
    "mx66": [("exact", "mx:e2m3:e2m3")] * 5,
    "mx64": [("exact", "mx:e2m3:e2m1")] * 5,
    "mx46": [("exact", "mx:e2m1:e2m3")] * 5,
    "mx88": [("exact", "mx:e4m3:e4m3")] * 5,
    "mx4_46": [("exact", "mx:e2m1:e2m1/e2m3")] * 5,
    "mx4_64": [("exact", "mx:e2m1:e2m3/e2m1")] * 5,
    "mx4_t6": [("exact", "mx:e2m1:trunc8/e2m3")] * 5,
    "mx4_t8": [("exact", "mx:e2m1:trunc8/e4m3")] * 5,
    "mx4_48": [("exact", "mx:e2m1:e2m1/e4m3")] * 5,
    "mx46d": [("exact", "mx:e2m1:e2m3:d")] * 5,
    "mx4_": [("exact", "mx:e2m1:none")] * 5,
    "mx_4": [("exact", "mx:none:e2m1")] * 5,
    "mx_46": [("exact", "mx:none:e2m1/e2m3")] * 5,          # f16mx without the activation-residual term
    # ... without it in some layers only (F = full f16mx, N = no activation-residual term)
    "mx_FNNFF": [("exact", "mx:e2m1:e2m1/e2m3")] + [("exact", "mx:none:e2m1/e2m3")] * 2 + [("exact", "mx:e2m1:e2m1/e2m3")] * 2,
    "mx_NNNFF": [("exact", "mx:none:e2m1/e2m3")] * 3 + [("exact", "mx:e2m1:e2m1/e2m3")] * 2,
    "mx_FFFNN": [("exact", "mx:e2m1:e2m1/e2m3")] * 3 + [("exact", "mx:none:e2m1/e2m3")] * 2,
    "mx_FNFFF": [("exact", "mx:e2m1:e2m1/e2m3")] + [("exact", "mx:none:e2m1/e2m3")] + [("exact", "mx:e2m1:e2m1/e2m3")] * 3,
    # ... round 6: which residual term does each layer need? (F = full; N = no activation-residual term x_l4 w_4; W = no weight-residual term x_4 w_l6)
    "mx_FFFFN": [("exact", "mx:e2m1:e2m1/e2m3")] * 4 + [("exact", "mx:none:e2m1/e2m3")],
    "mx_FFFNF": [("exact", "mx:e2m1:e2m1/e2m3")] * 3 + [("exact", "mx:none:e2m1/e2m3")] + [("exact", "mx:e2m1:e2m1/e2m3")],
    "mx_FFNFF": [("exact", "mx:e2m1:e2m1/e2m3")] * 2 + [("exact", "mx:none:e2m1/e2m3")] + [("exact", "mx:e2m1:e2m1/e2m3")] * 2,
    "mx_NFFFF": [("exact", "mx:none:e2m1/e2m3")] + [("exact", "mx:e2m1:e2m1/e2m3")] * 4,
    "mx_FFFFW": [("exact", "mx:e2m1:e2m1/e2m3")] * 4 + [("exact", "mx:e2m1:none")],
    "mx_FFFWF": [("exact", "mx:e2m1:e2m1/e2m3")] * 3 + [("exact", "mx:e2m1:none")] + [("exact", "mx:e2m1:e2m1/e2m3")],
    "mx_FFWFF": [("exact", "mx:e2m1:e2m1/e2m3")] * 2 + [("exact", "mx:e2m1:none")] + [("exact", "mx:e2m1:e2m1/e2m3")] * 2,
    "mx_FWFFF": [("exact", "mx:e2m1:e2m1/e2m3")] + [("exact", "mx:e2m1:none")] + [("exact", "mx:e2m1:e2m1/e2m3")] * 3,
    "mx_WFFFF": [("exact", "mx:e2m1:none")] + [("exact", "mx:e2m1:e2m1/e2m3")] * 4,
}

if __name__ == "__main__":
    names = sys.argv[1:] or ["two_pass", "one_plane_P0", "one_plane_P1", "one_plane_P2", "one_plane_P3", "one_plane_P4"]
    seeds = [int(s) for s in os.environ.get("SEEDS", "4321,1,2,3").split(",")]
    ins = inputs()
    feats = {k: features(v) for k, v in ins.items()}
    if os.environ.get("FRAMES"):                      # FRAMES=400,640: every input cut to its first n voiced frames (round 6: what each layer's residual terms are worth on short utterances)
        feats = {f"{k}[:{n}]": f[: int(n)] for k, f in feats.items() for n in os.environ["FRAMES"].split(",") if f.shape[0] >= int(n)}
    print("voiced frames:", {k: v.shape[0] for k, v in feats.items()}, flush=True)
    exact_s = [("exact", "exact")] * 5
    for seed in seeds:
        w = synth.make_weights(seed=seed)
        ex = {k: run(f, w, exact_s) for k, f in feats.items()}
        cal = None
        if any("h_cal" in (m for _, m in SCHEMES[n]) for n in names):
            col = {}
            for s_ in (777, 778):
                run(features(synth.make_wav(1, 160000, seed=s_)[0].astype(np.float64)), w, SCHEMES["two_pass"], collect=col)
            cal = {i: np.concatenate(v, 0).mean(0) for i, v in col.items()}
        for nm in names:
            errs = {k: np.abs(run(f, w, SCHEMES[nm], cal=cal) - ex[k]).max() for k, f in feats.items()}
            print(f"seed {seed:5d} {nm:16s} " + "  ".join(f"{k} {e * 1e5:6.2f}" for k, e in errs.items()), flush=True)
