#!/usr/bin/env python3
"""
bench.py — x-vectors/sec of the wav -> x-vector hot path on MI355X (BASELINE.json metric).

One step = one pass of the whole hot path (fused MFCC -> VAD/compaction/CMVN -> 5 TDNN GEMMs -> stats pooling ->
tdnn6 -> LDA/length-norm) over one batch of synthetic 10 s / 16 kHz utterances that is already resident in HBM.
Workload (config.workload): the BASELINE "8 192 utterances over 8 GPUs" configuration = 1 024 utterances per GPU
(weak scaling: per-GPU batch fixed), 0008_sitw_v2_1a topology with seeded random weights, dither 0.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line. `roofline` is for the dominant kernel (the TDNN MFMA GEMM launches, timed live with HIP
events on the launch stream); `cpu_baseline` times the NumPy oracle (a port of the reference's TF-CPU op graph) on a
bounded sample of the same workload on this box's host cores (and, as the checker, reports the GPU result's deviation
from it on two short utterances); nothing else in this file touches oracle/.
"""

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_FRAME_TDNN = 2 * 2_679_808          # 5 frame-level layers
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--gemm", default="bf16", choices=["bf16", "f16", "bf16x3", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=200)
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the fp32 / parity side measurements")
    args = ap.parse_args()

    import kaldi_tflite_amd as ktf
    from kaldi_tflite_amd import ops, parallel
    import synth

    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if os.environ.get("KTF_SHARE_GPU"):          # test hook: several ranks on one GPU (with KTF_DIST_BACKEND=gloo)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = synth.extractor_cfg(dither=0.0)
    w = synth.make_weights(seed=4321, narrow=False)
    mdl = synth.build_extractor(ktf, cfg, w, gemm=args.gemm)

    B, N = args.batch, int(args.seconds * 16000)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    wav = torch.clamp(torch.round(1000.0 * torch.randn((B, N), generator=g, device=dev)), -32767, 32767)
    T = mdl.framing.numFrames(N)

    def step():
        y = mdl(wav)
        if world > 1 and not args.no_gather:
            y = parallel.gather_embeddings(y, world)
        return y

    for _ in range(args.warmup):
        step()
    # a full Python GC pass over torch's ~170k long-lived objects stalls the host for 30-40 ms (measured: it landed in
    # one call of a side measurement and doubled its time): collect now and park the survivors in the permanent generation
    gc.collect()
    gc.freeze()
    # ---- timed region: exactly K steps between barrier + synchronize
    ops_prof = _GemmProfiler(ops)
    parallel.barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    parallel.barrier(world)
    torch.cuda.synchronize()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, world, dev)
    gemm_stats = ops_prof.finish()

    lens = mdl._ws[next(iter(mdl._ws))]["lens"].cpu().numpy()
    assert int(lens.min()) == T and int(lens.max()) == T, "synthetic stationary noise must keep every frame voiced"
    assert bool(torch.isfinite(y).all())

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    value = world * B * args.steps / dt
    out = {
        "metric": "x-vectors/sec (10 s @16 kHz)", "value": value, "unit": "x-vectors/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.gemm,
        "data": "synthetic",
        "config": {"workload": f"0008_sitw_v2_1a wav->x-vector, {args.seconds:g} s @16 kHz utterances, {B} per GPU "
                               f"(BASELINE config: 8192 utterances batch-sharded over 8 GPUs = 1024 per GPU), dither 0, all {T} frames voiced",
                   "utterances_per_gpu": B, "samples_per_utterance": N, "frames_per_utterance": T,
                   "tdnn_gemm": args.gemm, "weights": "synthetic seed 4321 (pretrained final.raw not shipped)",
                   "gather": bool(world > 1 and not args.no_gather)},
    }
    # ---- roofline of the dominant kernel: TDNN GEMM launches (5 per step), algorithmic FLOPs / measured duration
    flops_per_step = B * T * FLOP_PER_FRAME_TDNN
    gemm_ms_per_step = gemm_stats["total_ms"] / args.steps
    achieved = flops_per_step / (gemm_ms_per_step * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.gemm]
    out["roofline"] = {
        "bound": "mfma", "kernel": {"bf16": "tdnn_bf16r16_kernel (K=1536 layers) + tdnn_bf16h_kernel (K<=768 layers)", "bf16x3": "tdnn_x3r_kernel", "f32": "tdnn_f32_kernel",
                                       "f16": "tdnn_bf16r16_kernel<.., F16> + tdnn_bf16h_kernel<.., F16>"}[args.gemm],
        "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
        "launches_per_step": gemm_stats["launches"] // args.steps, "avg_launch_ms": gemm_stats["total_ms"] / max(gemm_stats["launches"], 1),
        "gemm_ms_per_step": gemm_ms_per_step, "per_layer_ms": gemm_stats["per_layer_ms"],
        "algorithmic_flop_per_step": flops_per_step,
        "per_layer_tflops": {k: _layer_flops(k, B, T) / (v * 1e-3) / 1e12 for k, v in gemm_stats["per_layer_ms"].items()},
        "note": "bf16x3 issues 3 MFMA passes per algorithmic FLOP" if args.gemm == "bf16x3" else "",
    }
    # HBM traffic of those launches comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE),
    # committed under profiles/: it cannot be collected from inside this process
    tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
    if args.gemm == "bf16" and B == 1024 and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        nl = max(out["roofline"]["launches_per_step"], 1)
        out["roofline"]["traffic"] = tj["tdnn_gemm_bytes_per_step_corrected"] / nl
        out["roofline"]["traffic_note"] = ("HBM bytes per launch = PMC bytes per step (profiles/r1_traffic.json: "
                                           f"{tj['tdnn_gemm_bytes_per_step_corrected']:.4g}) / {nl} GEMM launches; algorithmic "
                                           f"{tj['tdnn_gemm_algorithmic_bytes_per_step'] / nl:.4g} per launch")
    out["mfcc"] = _bench_mfcc(mdl, wav, ops)
    # side measurements and the CPU baseline belong to the single-GPU run only (rank 0 at N = 1)
    if not args.no_extra and world == 1:
        out["other_configs"] = _other_configs(ktf, synth, cfg, w, wav, args.gemm, dev)
    if not args.no_cpu_baseline and world == 1:
        # the only leg that touches oracle/: the CPU port timed as the baseline, and (as the checker) the deviation of the
        # GPU result from it on two short utterances
        out["cpu_baseline"] = _cpu_baseline(synth, cfg, w, args.cpu_utts, N)
        out["cpu_baseline"]["gpu_max_abs_dev_vs_fp64_oracle"] = _parity_sample(ktf, synth, cfg, w, args.gemm, dev)
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


class _GemmProfiler:
    """Brackets every ktf_tdnn launch of the frame-level layers with HIP events on the launch stream."""

    def __init__(self, ops):
        self.ops = ops
        self.orig = ops.tdnn
        self.events = []
        prof = self

        def wrapped(x, lens, desc, *a, **k):
            if x.shape[0] * x.shape[1] < 4096:          # tdnn6 (one row per utterance) is not the dominant kernel
                return prof.orig(x, lens, desc, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = prof.orig(x, lens, desc, *a, **k)
            e.record()
            prof.events.append(((int(desc.units), int(desc.nctx), int(desc.din)), s, e))
            return r

        self.orig_stats = ops.tdnn_stats

        def wrapped_stats(x, lens, desc, *a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = prof.orig_stats(x, lens, desc, *a, **k)
            e.record()
            prof.events.append(((int(desc.units), int(desc.nctx), int(desc.din), "+stats"), s, e))
            return r

        self.orig_split, self.orig_split_stats = ops.tdnn_split, ops.tdnn_split_stats

        def wrapped_split(xp, lens, desc, *a, **k):      # bf16x3 with hi/lo activation planes (xp: (2,B,T,ld))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = prof.orig_split(xp, lens, desc, *a, **k)
            e.record()
            prof.events.append(((int(desc.units), int(desc.nctx), int(desc.din)), s, e))
            return r

        def wrapped_split_stats(xp, lens, desc, *a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = prof.orig_split_stats(xp, lens, desc, *a, **k)
            e.record()
            prof.events.append(((int(desc.units), int(desc.nctx), int(desc.din), "+stats"), s, e))
            return r

        ops.tdnn = wrapped
        ops.tdnn_stats = wrapped_stats
        ops.tdnn_split = wrapped_split
        ops.tdnn_split_stats = wrapped_split_stats

    def finish(self):
        self.ops.tdnn = self.orig
        self.ops.tdnn_stats = self.orig_stats
        self.ops.tdnn_split, self.ops.tdnn_split_stats = self.orig_split, self.orig_split_stats
        total, per = 0.0, {}
        for key, s, e in self.events:
            ms = s.elapsed_time(e)
            total += ms
            name = f"{key[1]}x{key[2]}->{key[0]}" + (key[3] if len(key) > 3 else "")
            per[name] = per.get(name, 0.0) + ms
        n = max(len(self.events), 1)
        steps = max(len(self.events) // 5, 1)
        return {"total_ms": total, "launches": len(self.events), "per_layer_ms": {k: v / steps for k, v in per.items()}, "n": n}


def _bench_mfcc(mdl, wav, ops, iters=5):
    """Secondary BASELINE metric: MFCC frames/s per GPU (fused Framing+MFCC kernel alone), HBM roofline fraction."""
    from kaldi_tflite_amd import _lib as L
    B, N = wav.shape
    fr, mf = mdl.framing, mdl.mfcc
    T = fr.numFrames(N)
    cfg = L.FrontendCfg.from_buffer_copy(mf._cfg)
    cfg.frame_size, cfg.frame_shift = fr.frameWidth, fr.frameShift
    out = torch.empty((B, T, mf.numMfccs), dtype=torch.float32, device=wav.device)
    tabs = mf.tables(wav.device)
    ops.frontend(wav, L.IN_WAV, cfg, tabs, L.OUT_MFCC, N, B, T, out=out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.frontend(wav, L.IN_WAV, cfg, tabs, L.OUT_MFCC, N, B, T, out=out)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    frames = B * T
    gbs = frames * (fr.frameShift * 4 + mf.numMfccs * 4) / (ms * 1e-3) / 1e9
    return {"frames_per_s": frames / (ms * 1e-3), "ms": ms, "algorithmic_bytes_per_frame": fr.frameShift * 4 + mf.numMfccs * 4,
            "achieved_GBps": gbs, "hbm_peak_GBps": 8000.0, "frac_of_hbm_peak": gbs / 8000.0}


def _parity_sample(ktf, synth, cfg, w, gemm, dev):
    """max-abs deviation of the benchmarked configuration against the fp64 CPU oracle on 2 shorter utterances."""
    from oracle import ktf_oracle as O
    wav = synth.make_wav(2, 16000 * 3, seed=4242, ragged=True)
    want = O.xvector_forward(wav, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    res = {}
    for g in sorted({gemm, "f32", "f16", "bf16x3"}):
        m = synth.build_extractor(ktf, cfg, w, gemm=g)
        m.xvec.min_tiles = {}          # two short utterances would be routed to the fp32 kernels: measure the mode's own
        got = m(torch.as_tensor(wav, device=dev)).cpu().numpy()
        res[f"max_abs_dev_{g}"] = float(np.abs(got - want).max())
    return res


def _layer_flops(key, B, T):
    """FLOPs of all launches filed under a per_layer_ms key 'KxD->U[+stats]' (layers of equal shape share a key)."""
    kd, u = key.split("->")
    k, d = kd.split("x")
    n = {"3x512->512": 2}.get(key.replace("+stats", ""), 1)      # tdnn2 and tdnn3
    return 2.0 * B * T * int(k) * int(d) * int(u.replace("+stats", "")) * n


def _time_ms(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters


def _other_configs(ktf, synth, cfg, w, wav, gemm, dev):
    """Side measurements of the remaining BASELINE.json configs on the same GPU (not part of `value`):
    the same 1024-utterance step in the other GEMM arithmetic modes, batch-1 latency (config 2) and the
    1024 x 1024 PLDA trial matrix (config 5)."""
    res = {}
    B = wav.shape[0]
    for g in ("f16", "bf16x3", "f32"):
        if g == gemm:
            continue
        m = synth.build_extractor(ktf, cfg, w, gemm=g)
        ms = _time_ms(lambda: m(wav), 2)
        res[f"x_vectors_per_s_{g}"] = B / (ms * 1e-3)
        del m
        torch.cuda.empty_cache()
    # int16 PCM input: same step, half the input bytes; and the PCIe-inclusive rate of a host-fed step
    mi = synth.build_extractor(ktf, cfg, w, gemm=gemm)
    wav16 = wav.to(torch.int16)
    res[f"x_vectors_per_s_{gemm}_int16_input"] = B / (_time_ms(lambda: mi(wav16), 3) * 1e-3)
    host16 = wav16.cpu().pin_memory()
    res[f"x_vectors_per_s_{gemm}_int16_from_pinned_host"] = B / (_time_ms(lambda: mi(host16.to(dev, non_blocking=True)), 3) * 1e-3)
    hb = [host16] * 8

    def streamed():
        for y in mi.extract_stream(hb, depth=3):
            pass
    res[f"x_vectors_per_s_{gemm}_int16_from_pinned_host_overlapped"] = 8 * B / (_time_ms(streamed, 2) * 1e-3)   # 8 batches, first upload exposed
    del mi, wav16, host16, hb
    torch.cuda.empty_cache()
    m1 = synth.build_extractor(ktf, cfg, w, gemm="f32")
    one = wav[:1].contiguous()
    res["batch1_fp32_latency_ms"] = _time_ms(lambda: m1(one), 10)
    rng = np.random.default_rng(31)
    dim, nb = 128, 1024
    A = rng.standard_normal((dim, dim)) / np.sqrt(dim) + np.eye(dim)
    plda = ktf.layers.PLDA(dim, rng.standard_normal(dim) * 0.1, A, np.sort(rng.uniform(0.05, 30.0, dim))[::-1].copy())
    xv = torch.as_tensor(rng.standard_normal((nb, dim)), device=dev)
    res["plda_1024x1024_fp64_ms"] = _time_ms(lambda: plda(xv), 5)
    return res


def _cpu_baseline(synth, cfg, w, n_utts, N):
    """The NumPy oracle (a port of the reference's TF-CPU op graph: materialised frames, rfft, dense mel matmul,
    materialised im2col + GEMM) on `n_utts` utterances of the same workload, fp32, all host cores via BLAS threads."""
    from oracle import ktf_oracle as O
    layers = synth.oracle_layers(w)
    wav = synth.make_wav(1, N, seed=1)
    O.xvector_forward(wav, cfg, layers, w["mean"], w["lda"], dtype=np.float32)     # warm-up
    wav = synth.make_wav(n_utts, N, seed=2)
    t0 = time.perf_counter()
    O.xvector_forward(wav, cfg, layers, w["mean"], w["lda"], dtype=np.float32)
    dt = time.perf_counter() - t0
    return {"value": n_utts / dt, "unit": "x-vectors/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_utts} utterances of the same 10 s workload, one at a time (the reference is batch-1 only), "
                      f"NumPy fp32 oracle, {dt:.1f} s of CPU work"}


if __name__ == "__main__":
    main()
