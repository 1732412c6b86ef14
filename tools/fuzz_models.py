#!/usr/bin/env python3
"""Differential fuzzing of the Sequential runner against the oracle (GPU box): random TDNN stacks (units and input widths off the tile
sizes, VALID padding, subsampling, fused / own activations, BatchNorm, an optional reducing StatsPooling with layers behind it) on
ragged batches of random size, in every arithmetic mode, default routing (small-batch tiles, pair route, planes, flat rows, loader
kernel). Test infrastructure: the oracle is the checker.   python tools/fuzz_models.py [n_cases] [seed] [--knobs] [--big] [--tiny]"""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import kaldi_tflite_amd as ktf
from oracle import ktf_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
BIG = "--big" in sys.argv       # batches of hundreds of utterances: the checker is this library's own fp32 mode (checked against the oracle at the
                                # small sizes), the oracle would take minutes per case
TOL = {"f32": 2e-5, "bf16x3": 3e-4, "f16mx": 4e-3, "bf16": 1.5e-1}       # relative to the output's largest magnitude
bad = 0
for case in range(n_cases):
    gemm = str(rng.choice(["f32", "bf16x3", "f16mx", "f16mx", "bf16"])) if not BIG else str(rng.choice(["bf16x3", "f16mx", "f16mx", "bf16"]))
    D = int(rng.choice([24, 30, 40, 64, 100]))
    n_frame = int(rng.integers(1, 5))
    pooled_at = n_frame if rng.random() < 0.5 else None
    n_post = int(rng.integers(0, 3)) if pooled_at is not None else 0
    spec, lcfg = [], [{"name": "input", "type": "input", "shape": [None, None, D]}]
    for i in range(n_frame + n_post):
        post = i >= n_frame
        U = int(rng.choice([16, 96, 130, 256, 300, 512]))
        ctx = [0] if post else sorted(set(int(v) for v in rng.integers(-4, 5, int(rng.integers(1, 4)))))
        pad = "SAME" if post or rng.random() < 0.7 else "VALID"
        sub = 1 if post or rng.random() < 0.75 else int(rng.choice([2, 3]))
        form = str(rng.choice(["affine", "affine+relu+bn", "affine+relu", "own_relu", "own_tanh"]))
        spec.append((U, ctx, pad, sub, form))
        kinds = {"affine": "affine", "affine+relu+bn": ["affine", "relu", "batchnorm"], "affine+relu": ["affine", "relu"]}.get(form, "affine")
        c = {"units": U, "context": ctx, "padding": pad, "subsampling_factor": sub}
        if form.startswith("own_"):
            c["activation"] = form[4:]
        lcfg.append({"name": f"t{i}", "type": kinds, "cfg": c})
        if pooled_at is not None and i + 1 == n_frame:
            lcfg.append({"name": "stats", "type": "stats", "cfg": {"left_context": 0, "right_context": 10000, "reduce_time_axis": True,
                                                                     "include_std": bool(rng.integers(0, 2))}})
    # a third of the small cases go through Sequential.__call__ on a dense batch instead of the ragged runner, some of them with a WINDOWED
    # StatsPooling behind the last frame-level layer (the fused runner hands such stacks to the layer-by-layer path)
    dense = (not BIG) and rng.random() < 0.33
    win_stats = None
    if dense and pooled_at is None and rng.random() < 0.6:
        ip = int(rng.choice([1, 1, 2]))
        win_stats = {"left_context": -int(rng.integers(0, 6)), "right_context": int(rng.integers(0, 6)), "input_period": ip,
                     "output_period": ip * int(rng.choice([1, 2])), "include_std": bool(rng.integers(0, 2)), "padding": str(rng.choice(["SAME", "VALID"]))}
        lcfg.append({"name": "wstats", "type": "stats", "cfg": win_stats})
    try:
        mdl = ktf.models.SequentialFromConfig({"type": "sequential", "layers": lcfg}, None, "m", gemm=gemm)
    except Exception as e:
        print(f"SKIP (config) {lcfg}: {e}")
        continue
    knobs = {}
    if "--knobs" in sys.argv:                            # random settings of the runner's A/B knobs: schedules and kernels change, the contract does not
        knobs = {"mx_loader": [None, True, False][int(rng.integers(0, 3))], "flat_rows": bool(rng.integers(0, 2)), "mx_flat_rows": bool(rng.integers(0, 2)), "flat_pooling": bool(rng.integers(0, 2)),
                 "split_planes": bool(rng.integers(0, 2)), "fuse_stats": bool(rng.integers(0, 2)), "deterministic": bool(rng.integers(0, 2)),
                 "small_tile_pairs": bool(rng.integers(0, 2))}
        if os.environ.get("FUZZ_FORCE"):                # "knob=value,...": pin knobs after the draw (same random sequence): bisecting a mismatch
            for kv in os.environ["FUZZ_FORCE"].split(","):
                k, v = kv.split("=")
                knobs[k] = {"True": True, "False": False, "None": None}[v]
        for k, v in knobs.items():
            setattr(mdl, k, v)
        if rng.random() < 0.5:
            mdl.min_tiles, knobs["min_tiles"] = {}, {}
        if rng.random() < 0.5:
            mdl.min_frames, knobs["min_frames"] = {}, {}
    layers, din = [], D
    for i, (U, ctx, pad, sub, form) in enumerate(spec):
        W = (rng.standard_normal((U, len(ctx) * din)) / np.sqrt(len(ctx) * din)).astype(np.float32)
        b = (rng.standard_normal(U) * 0.1).astype(np.float32)
        mdl.get_layer(f"t{i}.affine").set_weights([W, b])
        L = {"kind": "tdnn", "W": W, "b": b, "context": ctx, "padding": pad, "subsampling_factor": sub}
        if form.startswith("own_"):
            L["activation"] = form[4:]
        layers.append(L)
        if "relu" in form and not form.startswith("own_"):
            layers.append({"kind": "relu"})
        if form.endswith("bn"):
            bn = (np.float32(1.0), rng.uniform(-0.2, 0.4, U).astype(np.float32), rng.uniform(0.5, 2.0, U).astype(np.float32))
            mdl.get_layer(f"t{i}.batchnorm").set_weights(list(bn))
            layers.append({"kind": "bn", "rms": bn[0], "mean": bn[1], "var": bn[2]})
        din = U
        if pooled_at is not None and i + 1 == n_frame:
            sc = [e for e in lcfg if e["name"] == "stats"][0]["cfg"]
            layers.append({"kind": "stats", **sc})
            din = 2 * U if sc["include_std"] else U
            # the layer behind the pooling was built for din: rebuild the weight shapes lazily below
    # widths behind the pooling: the config builder sized them from the pooled width; regenerate consistent weights
    if win_stats is not None:
        layers.append({"kind": "stats", **win_stats})
    B = int(rng.choice([1, 2, 3, 7, 40])) if not BIG else int(rng.choice([130, 257, 600, 1024]))
    T = int(rng.choice([12, 40, 150, 300, 700])) if not BIG else int(rng.choice([100, 257, 998]))
    lens = rng.integers(max(1, T // 3), T + 1, B).astype(np.int32)
    lens[int(rng.integers(0, B))] = T
    if "--tiny" in sys.argv and B > 1 and rng.random() < 0.6:       # utterances of a few frames: VALID-padded layers leave none of them (NaN statistics, as in
        for _ in range(int(rng.integers(1, 3))):                     # the reference), SAME-padded ones replicate their edges across whole contexts
            lens[int(rng.integers(1, B))] = int(rng.integers(1, 7))
        lens[0] = T
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    if dense:
        lens[:] = T
    desc = dict(gemm=gemm, D=D, B=B, T=T, lens=lens.tolist() if not dense else "dense", spec=spec, pooled_at=pooled_at, win_stats=win_stats, knobs=knobs)
    try:
        if dense:
            got = mdl(torch.as_tensor(x, device="cuda")).float().cpu().numpy()
        else:
            got = mdl.run_ragged(torch.as_tensor(x, device="cuda"), torch.as_tensor(lens, device="cuda")).float().cpu().numpy()
    except NotImplementedError:
        continue
    except Exception as e:
        bad += 1
        print(f"MISMATCH {desc}: runner raises {type(e).__name__}: {e}", flush=True)
        continue
    if BIG:
        ref = ktf.models.SequentialFromConfig({"type": "sequential", "layers": lcfg}, None, "m32", gemm="f32")
        ti = 0
        for L in layers:                                   # the same Kaldi-format weights as the model under test
            if L["kind"] == "tdnn":
                ref.get_layer(f"t{ti}.affine").set_weights([L["W"], L["b"]])
                ti += 1
            elif L["kind"] == "bn":
                ref.get_layer(f"t{ti - 1}.batchnorm").set_weights([L["rms"], L["mean"], L["var"]])
        ref_out = ref.run_ragged(torch.as_tensor(x, device="cuda"), torch.as_tensor(lens, device="cuda")).float().cpu().numpy()
        def n_out(n):
            for L in layers:
                if L["kind"] == "tdnn":
                    n = O.tdnn_eval_indices(n, L["context"], L.get("subsampling_factor", 1), L.get("padding", "SAME")).shape[0] if n > 0 else 0
                elif L["kind"] == "stats":
                    n = 1
            return n
        out_lens = [n_out(int(n)) for n in lens]
    for b in range(B):
        try:
            want = ref_out[b, : out_lens[b]].astype(np.float64) if BIG else O.sequential_forward(layers, x[b:b + 1, : lens[b]], dtype=np.float64)[0]
        except Exception as e:
            print(f"SKIP (oracle) {desc}: {e}")
            break
        n = want.shape[0]
        g = got[b, :n] if got.shape[1] >= n else None
        if g is None or g.shape != want.shape:
            bad += 1
            print(f"MISMATCH {desc} utterance {b}: output {got.shape} for {want.shape}", flush=True)
            break
        if n == 0:
            continue
        okw = np.isfinite(want)
        if not np.array_equal(okw, np.isfinite(g)):
            bad += 1
            print(f"MISMATCH {desc} utterance {b}: non-finite pattern differs (oracle finite {int(okw.sum())} of {okw.size}, runner finite {int(np.isfinite(g).sum())}; "
                  f"runner values {g.ravel()[:4]}, oracle {want.ravel()[:4]})", flush=True)
            break
        err = np.abs(g[okw] - want[okw]).max() / max(1e-6, np.abs(want[okw]).max()) if okw.any() else 0.0
        if not err <= TOL[gemm]:
            bad += 1
            print(f"MISMATCH {desc} utterance {b}: relative deviation {err:.3e} > {TOL[gemm]}", flush=True)
            break
print(f"{n_cases} rounds, {bad} mismatches")
sys.exit(min(bad, 255))
