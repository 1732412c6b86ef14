#!/usr/bin/env python3
"""
bench.py — x-vectors/sec of the wav -> x-vector hot path on MI355X (BASELINE.json metric).

One step = one pass of the whole hot path (fused MFCC -> VAD/compaction/CMVN -> 5 TDNN GEMMs with fused pooling ->
tdnn6 -> LDA/length-norm) over one batch of synthetic 10 s / 16 kHz utterances that is already resident in HBM.
Workload (config.workload): the BASELINE "8 192 utterances over 8 GPUs" configuration = 1 024 utterances per GPU
(weak scaling: per-GPU batch fixed), 0008_sitw_v2_1a topology with seeded random weights, dither 0.

The timed arithmetic (`dtype`) defaults to "f16mx" (csrc/tdnn_mx.hip): ONE half-precision MFMA pass plus two block-scaled
(OCP MX) residual passes at four times the half rate on v_mfma_scale_f32_16x16x128_f8f6f4 -- x_h w_h + fp4(x - x_h) fp4(w) +
fp4(x_h) fp6(w - w_h), one power-of-two scale per 32 K elements -- 1.5 MFMA passes per algorithmic flop with ~15 significant
bits on both operands and NO calibration. It is the fastest mode that meets north_star's <= 1e-4 max-abs deviation ON EVERY
INPUT: the line's `max_abs_dev_vs_fp64_oracle` is the maximum over stationary noise (the timed workload), utterances with quiet
blocks AND the reference's own end-to-end speech recording (whole and in 10 s chunks), and `timed_batch_vs_f32` compares all
1024 timed x-vectors with the exact fp32 kernels. `--gemm bf16x3` (split-bf16) and `--gemm f32` (exact) are the tighter
modes (bf16x3 carries its own roofline block in `other_configs`); one-pass `--gemm bf16` is outside the tolerance (reported
with tolerance_ok false).

    python bench.py                      # 1 GPU
    python bench.py --gpus 8             # starts 8 ranks itself (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line. `roofline` is for the dominant kernel (the TDNN MFMA GEMM launches, timed live with HIP
events on the launch stream); `cpu_baseline` times the torch-CPU restatement of the reference's op graph
(oracle/ktf_torch_cpu.py) on this box's host cores and, as the checker, the NumPy fp64 oracle gives the deviation of the GPU
result; nothing else in this file touches oracle/.
"""

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "kaldi-tflite_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOP_PER_FRAME_TDNN = 2 * 2_679_808          # 5 frame-level layers
MFCC_FLOP_PER_FRAME = 25_000                 # FFT-512 + window + sparse mel + DCT
PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0, "f16mx": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks
MFMA_PASSES = {"bf16": 1, "bf16x3": 3, "f16mx": 1.5, "f32": 1}                        # MFMA passes per algorithmic FLOP
TOLERANCE = 1e-4                             # north_star: max-abs x-vector deviation vs the fp32 reference path
KERNELS = {"bf16": "tdnn_bf16r16_kernel (K=1536 layers) + tdnn_bf16h_kernel (K<=768 layers)",
           "bf16x3": "tdnn_x3r_kernel<.., SPLIT> (tdnn2-4) + tdnn_x3s_kernel (tdnn5 + pooling) + tdnn_x3r_kernel (tdnn1)",
           "f16mx": "tdnn_mx_kernel (csrc/tdnn_mx.hip): v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 (fp4 x fp4, fp4 x fp6) per 128 K",
           "f32": "tdnn_f32t_kernel"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the extra region that runs beside the one-wave shader-clock probe (profiling runs: "
                                                                  "it would top the kernel statistics with its own duration); shader_clock_mhz is then null")
    ap.add_argument("--no-mx-flat", action="store_true", help="A/B: f16mx plane layers on per-utterance 256-row tiles instead of flat row tiles")
    ap.add_argument("--repeats", type=int, default=3, help="the timed region (exactly --steps steps between barrier + synchronize) is run this many "
                                                           "times back to back; `value` / `ms_per_step` are the MEDIAN region, `value_runs` lists all")
    ap.add_argument("--batch", type=int, default=1024, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--gemm", default="f16mx", choices=["bf16", "bf16x3", "f16mx", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--dump-xvectors", default=None, help="rank 0 saves the x-vectors of the last timed step (all ranks' when gathered) as .npy (tests)")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (other modes, latency, PLDA)")
    ap.add_argument("--no-parity", action="store_true", help="measurement runs of timing-only ablation builds (tools/mx): no oracle comparison, no finite check")
    ap.add_argument("--mx-loader", action="store_true", help="A/B: f16mx on the loader-wave kernel (csrc/tdnn_mxl.hip) instead of the 256 x 256 eight-wave kernel")
    ap.add_argument("--no-short-routing", action="store_true", help="A/B: without the device-side second pass over utterances below MIN_FRAMES voiced frames")
    ap.add_argument("--atomic-pooling", action="store_true", help="fp64-atomic fused pooling instead of the reproducible form")
    ap.add_argument("--n1-json", default=None, help="a file holding the JSON line of the N = 1 run of this bench: the N > 1 line then carries "
                                                    "`scaling_summary` (value / (N x value at N = 1)); the driver computes efficiency itself, this is a convenience")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------- self-launch (N > 1)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n, argv, port=None, python=None, script=None):
    """The `torch.distributed.run` command line that runs this file (or `script`) on n ranks of one node, one rank per GPU."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script or os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child job and pass its output through.
    Runs BEFORE this process touches the GPU (a process that has initialised HIP must not exec or share its context)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    # (host threads per rank: the ranks share the node's CPUs -- eight ranks x eight threads oversubscribed a 16-CPU lease)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // max(args.gpus, 1)))))
    cmd = launch_command(args.gpus, argv)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))

    import numpy as np  # noqa: F401
    import torch

    import kaldi_tflite_amd as ktf
    from kaldi_tflite_amd import ops, parallel
    import synth

    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if os.environ.get("KTF_SHARE_GPU"):          # test hook: several ranks on one GPU (with KTF_DIST_BACKEND=gloo)
        local_rank %= torch.cuda.device_count()
    else:                                        # one rank per GPU: two ranks on one device would halve both and still "scale"
        assert torch.cuda.device_count() >= world, f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = synth.extractor_cfg(dither=0.0)
    w = synth.make_weights(seed=4321, narrow=False)
    mdl = synth.build_extractor(ktf, cfg, w, gemm=args.gemm)
    mdl.xvec.deterministic = not args.atomic_pooling
    mdl.xvec.mx_flat_rows = not args.no_mx_flat
    mdl.xvec.mx_loader = True if args.mx_loader else None          # (None: the model picks per batch; 1024 x 10 s takes the 256-row kernel)
    mdl.route_short_utterances = not args.no_short_routing

    B, N = args.batch, int(args.seconds * 16000)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    wav = torch.clamp(torch.round(1000.0 * torch.randn((B, N), generator=g, device=dev)), -32767, 32767)
    T = mdl.framing.numFrames(N)

    def step():
        y = mdl(wav)
        if world > 1 and not args.no_gather:
            y = parallel.gather_embeddings(y, world)
        return y

    # a full Python GC pass over torch's ~170k long-lived objects stalls the host for 30-40 ms (measured: it landed in
    # one call of a side measurement and doubled its time): collect now and park the survivors in the permanent generation.
    # Before the warm-up, not after it: an idle gap in front of the timed region sends the GPU back down its clock ramp
    # (tools/ramp_probe.py: the first five steps after idling run 11.4 / 10.4 / 10.2 / 10.1 / 10.0 ms against 9.9 sustained).
    gc.collect()
    gc.freeze()
    ops_prof = _GemmProfiler(ops, torch)             # warm-up steps run through the same event-bracketed launches
    tw = time.perf_counter()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    est_us = 1e6 * (time.perf_counter() - tw) / max(args.warmup, 1) * args.steps       # (what one timed region will take, about)
    # ---- timed region: exactly K steps between barrier + synchronize; run `repeats` times back to back, the MEDIAN region is reported
    # (one 0.2 s window could not tell a 3 % kernel change from a slow box). Beside every region a one-wave probe on a
    # stream of its own reads the shader clock the chip holds under this load (ktf_clock_probe).
    # Round 6: the probe runs beside ONE MORE region of the same steps behind the timed ones, not beside them: a second active hardware
    # queue stretches every kernel-to-kernel transition of the step (same box, same build, alternating runs: 8.58 ms of GEMM brackets per
    # step with the probe resident, 8.40 without -- 2 % of the step, the pooled layer's bracket most), so the timed regions are measured
    # alone and the clock is sampled under the same load right after them (`shader_clock_region_ms_per_step` says what that region took).
    probe_stream = torch.cuda.Stream(device=dev)
    probe_out = torch.zeros((1, 4), dtype=torch.int64, device=dev)
    regions = []
    probe_region_ms = None
    n_regions = max(args.repeats, 1)
    for rep_i in range(n_regions + (0 if args.no_clock_probe else 1)):
        probing = rep_i == n_regions
        ops_prof.reset()
        parallel.barrier(world)
        torch.cuda.synchronize()
        if probing:
            ops.clock_probe(probe_out[0], max(1000, min(int(0.9 * est_us), 9_000_000)), probe_stream)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step()
        parallel.barrier(world)
        torch.cuda.current_stream().synchronize()
        dt_local = time.perf_counter() - t0
        torch.cuda.synchronize()                     # (the probe wave: it ends inside the region by construction, 0.9 x its estimate)
        dt_i = parallel.max_over_ranks(dt_local, world, dev)
        if probing:
            probe_region_ms = 1e3 * dt_i / args.steps
        else:
            regions.append({"dt": dt_i, "dt_local": dt_local, "gemm": ops_prof.finish(restore=False)})
    ops_prof.restore()
    order = sorted(range(len(regions)), key=lambda i: regions[i]["dt"])
    med = order[len(order) // 2]
    dt, gemm_stats = regions[med]["dt"], regions[med]["gemm"]
    per_rank = parallel.gather_floats(regions[med]["dt_local"], world, dev)         # every rank's own time of the median region
    pc = probe_out.cpu().numpy()
    clock_mhz = float(100.0 * pc[0][0] / pc[0][1]) if pc[0][1] else None

    lens = mdl.last_lens.cpu().numpy()
    assert int(lens.min()) == T and int(lens.max()) == T, "synthetic stationary noise must keep every frame voiced"
    assert args.no_parity or bool(torch.isfinite(y).all())
    ranks_seen = torch.distributed.get_world_size() if world > 1 else 1
    backend = torch.distributed.get_backend() if world > 1 else None

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.dump_xvectors:
        np.save(args.dump_xvectors, y.detach().cpu().numpy())

    value = world * B * args.steps / dt
    out = {
        "metric": "x-vectors/sec (10 s @16 kHz)", "value": value, "unit": "x-vectors/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.gemm,
        "data": "synthetic", "library": os.path.relpath(ktf._lib.LIB_PATH, ROOT), "library_build_id": ops.build_id(),
        # every timed region of this run (each exactly `steps` steps, max over ranks); `value` is the median one
        "repeats": len(regions), "value_runs": [world * B * args.steps / r["dt"] for r in regions],
        "ms_per_step_runs": [1e3 * r["dt"] / args.steps for r in regions],
        # shader clock held during each region (rank 0's device): in-kernel s_memtime against the 100 MHz s_memrealtime
        "shader_clock_mhz": clock_mhz, "shader_clock_region_ms_per_step": probe_region_ms if clock_mhz else None,
        "shader_clock_mhz_1ms_window_min_max": [float(pc[0][2]) / 1e3, float(pc[0][3]) / 1e3],
        "per_rank_ms_per_step": {"min": 1e3 * min(per_rank) / args.steps, "max": 1e3 * max(per_rank) / args.steps, "ranks": len(per_rank)},
        "config": {"workload": f"0008_sitw_v2_1a wav->x-vector, {args.seconds:g} s @16 kHz utterances, {B} per GPU "
                               f"(BASELINE config: 8192 utterances batch-sharded over 8 GPUs = 1024 per GPU), dither 0, all {T} frames voiced",
                   "utterances_per_gpu": B, "samples_per_utterance": N, "frames_per_utterance": T,
                   "tdnn_gemm": args.gemm,
                   "weights": "synthetic seed 4321 (pretrained final.raw not shipped)",
                   "gather": bool(world > 1 and not args.no_gather), "ranks_seen_by_collective_backend": ranks_seen,
                   "collective_backend": backend, "fused_pooling": "atomic" if args.atomic_pooling else "reproducible"},
    }
    # ---- roofline of the dominant kernel: TDNN GEMM launches (5 per step), algorithmic FLOPs / measured duration
    passes = float(MFMA_PASSES[args.gemm])
    out["roofline"] = _roofline(args.gemm, gemm_stats, args.steps, B, T, passes)
    if clock_mhz:
        # the dense-MFMA peak is quoted at the 2.4 GHz maximum clock; the governor holds less under this load, by amounts that differ from
        # device to device. The probe wave reads the clock of ITS OWN CU, which does nothing else: the CUs inside the GEMM K-loops hold
        # 2.0-2.1 GHz by their own stamps (docs/lab_notes_r5.md section 4), so this figure under-corrects -- it separates boxes, it is not
        # the clock of the matrix pipes (named for what it is)
        out["roofline"]["probe_cu_clock_mhz"] = clock_mhz
        out["roofline"]["peak_at_probe_cu_clock"] = out["roofline"]["peak"] * clock_mhz / 2400.0
        out["roofline"]["frac_at_probe_cu_clock"] = out["roofline"]["achieved"] / out["roofline"]["peak_at_probe_cu_clock"]
    # (flat scalars beside the nested blocks: a record that keeps only the scalar members of `roofline` / `config` still carries them)
    out["config"]["shader_clock_mhz_probe_cu"] = clock_mhz
    out["config"]["library_build_id"] = ops.build_id()
    for k_, v_ in out["roofline"]["per_layer_ms"].items():
        out["roofline"]["ms_" + k_.replace("->", "_to_").replace("+", "_")] = v_
    if args.gemm == "f16mx" and args.mx_loader:
        out["roofline"]["kernel"] = "tdnn_mxl_kernel (csrc/tdnn_mxl.hip: 192 x 256 tile, 8 matrix + 4 loader waves; --mx-loader A/B)"
    # HBM traffic of those launches comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE),
    # committed under profiles/: it cannot be collected from inside this process
    if B == 1024:
        _attach_traffic(out["roofline"], args.gemm, ops.build_id())
    out["mfcc"] = _bench_mfcc(torch, mdl, wav, ops)
    # the second half of BASELINE.json's metric ("MFCC frames/sec per GPU") as scalars of the roofline block too: the fused front-end launch alone,
    # against its two ceilings (760 algorithmic bytes per frame of HBM; ~25 kFLOP per frame of fp32 vector work)
    for k_ in ("frames_per_s", "ms", "frac_of_hbm_peak", "frac_of_valu_peak", "achieved_GBps"):
        out["roofline"]["frontend_" + k_] = out["mfcc"][k_]
    out["roofline"]["frontend_kernel"] = "frontend512_kernel (csrc/frontend512.hip): Framing + Windowing + FFT-512 + mel + DCT + lifter, one wave per frame"
    if args.n1_json:
        out["scaling_summary"] = _scaling_summary(args.n1_json, value, world)
    # side measurements and the CPU baseline belong to the single-GPU run only (rank 0 at N = 1)
    if world == 1 and not args.no_parity:
        # the deviation of the TIMED mode from the fp64 CPU oracle at the full utterance length: part of the headline
        modes = [args.gemm] if args.no_extra else sorted({"f32", "bf16x3", "f16mx", "bf16", args.gemm})
        dev_info = _parity_sample(torch, ktf, synth, cfg, w, modes, dev, N)
        out["max_abs_dev_vs_fp64_oracle"] = dev_info[args.gemm]["max"]
        out["max_abs_dev_by_input"] = dev_info[args.gemm]
        out["tolerance"] = TOLERANCE
        out["tolerance_ok"] = bool(dev_info[args.gemm]["max"] <= TOLERANCE)
        out["parity_sample"] = dev_info["sample"]
        out["timed_batch_vs_f32"] = _timed_batch_vs_f32(torch, ktf, synth, cfg, w, wav, y if world == 1 else mdl(wav))
        if not args.no_extra:
            out["other_configs"] = _other_configs(torch, ktf, synth, cfg, w, wav, args.gemm, dev, dev_info, mdl)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = _cpu_baseline(torch, synth, cfg, w, N)
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


class _GemmProfiler:
    """Brackets every ktf_tdnn launch of the frame-level layers with HIP events on the launch stream."""

    NAMES = ("tdnn", "tdnn_stats", "tdnn_split", "tdnn_split_stats", "tdnn_split_flat", "tdnn_split_flat_stats", "tdnn_mx", "tdnn_mx_flat", "tdnn_mx_stats",
             "tdnn_mx_flat_stats")
    AUX = ("mx_planes", "split_bf16")      # conversions a mode needs in front of its first GEMM: timed too, reported separately

    def __init__(self, ops, torch):
        self.ops, self.torch = ops, torch
        self.events = []
        self.orig = {n: getattr(ops, n) for n in self.NAMES + self.AUX}
        self.aux_events = []
        for n in self.NAMES:
            setattr(ops, n, self._wrap(n))
        for n in self.AUX:
            setattr(ops, n, self._wrap_aux(n))

    def _wrap(self, name):
        orig, prof, torch = self.orig[name], self, self.torch
        split, stats = "split" in name, "stats" in name

        def wrapped(x, lens, desc, *a, **k):
            rows = (x.shape[1] * x.shape[2]) if (hasattr(x, "dim") and x.dim() == 4) else (x.shape[0] * x.shape[1])
            if rows < 4096:          # tdnn6 (one row per utterance) is not the dominant kernel
                return orig(x, lens, desc, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(x, lens, desc, *a, **k)
            e.record()
            prof.events.append((f"{int(desc.nctx)}x{int(desc.din)}->{int(desc.units)}" + ("+stats" if stats else ""), s, e))
            return r

        return wrapped

    def _wrap_aux(self, name):
        orig, prof, torch = self.orig[name], self, self.torch

        def wrapped(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **k)
            e.record()
            prof.aux_events.append((name, s, e))
            return r

        return wrapped

    def reset(self):
        """Drops the launches recorded so far (the warm-up)."""
        self.events = []
        self.aux_events = []

    def restore(self):
        for n in self.NAMES + self.AUX:
            setattr(self.ops, n, self.orig[n])

    def finish(self, restore=True):
        if restore:
            self.restore()
        aux = {}
        for name, s, e in self.aux_events:
            aux[name] = aux.get(name, 0.0) + s.elapsed_time(e)
        total, per = 0.0, {}
        for name, s, e in self.events:
            ms = s.elapsed_time(e)
            total += ms
            per[name] = per.get(name, 0.0) + ms
        if len(self.events) % 5:
            raise RuntimeError(f"bench: {len(self.events)} frame-level GEMM launches recorded, not a multiple of the five layers -- an ops entry point "
                               "is missing from _GemmProfiler.NAMES (the roofline would leave a layer out)")
        steps = max(len(self.events) // 5, 1)
        return {"total_ms": total, "launches": len(self.events), "per_layer_ms": {k: v / steps for k, v in per.items()},
                "aux_ms_per_step": {k: v / steps for k, v in aux.items()}}



def _bench_mfcc(torch, mdl, wav, ops, iters=20, warm=5):
    """Secondary BASELINE metric: MFCC frames/s per GPU (fused Framing+MFCC kernel alone) against both of its ceilings:
    HBM (760 algorithmic bytes per frame) and VALU (~25 kFLOP per frame at the fp32 vector peak)."""
    from kaldi_tflite_amd import _lib as L
    B, N = wav.shape
    fr, mf = mdl.framing, mdl.mfcc
    T = fr.numFrames(N)
    cfg = L.FrontendCfg.from_buffer_copy(mf._cfg)
    cfg.frame_size, cfg.frame_shift = fr.frameWidth, fr.frameShift
    out = torch.empty((B, T, mf.numMfccs), dtype=torch.float32, device=wav.device)
    tabs = mf.tables(wav.device)
    t0 = time.perf_counter()       # (the clock takes ~50 ms of launches to settle after an idle gap: 0.75 ms for the first 25 launches,
    while time.perf_counter() - t0 < 0.12:                              # 0.65 ms from the 50th on -- what the launch takes inside the step)
        for _ in range(max(warm, 1)):
            ops.frontend(wav, L.IN_WAV, cfg, tabs, L.OUT_MFCC, N, B, T, out=out)
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.frontend(wav, L.IN_WAV, cfg, tabs, L.OUT_MFCC, N, B, T, out=out)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    frames = B * T
    fps = frames / (ms * 1e-3)
    gbs = fps * (fr.frameShift * 4 + mf.numMfccs * 4) / 1e9
    return {"frames_per_s": fps, "ms": ms, "algorithmic_bytes_per_frame": fr.frameShift * 4 + mf.numMfccs * 4,
            "achieved_GBps": gbs, "hbm_peak_GBps": 8000.0, "frac_of_hbm_peak": gbs / 8000.0,
            "valu_flop_per_frame": MFCC_FLOP_PER_FRAME, "achieved_valu_TFLOPs": fps * MFCC_FLOP_PER_FRAME / 1e12,
            "valu_peak_TFLOPs": PEAK_TFLOPS["f32"], "frac_of_valu_peak": fps * MFCC_FLOP_PER_FRAME / 1e12 / PEAK_TFLOPS["f32"]}


def _parity_sample(torch, ktf, synth, cfg, w, modes, dev, N):
    """max-abs deviation from the fp64 CPU oracle (the checker), per GEMM mode, over three kinds of input at the FULL utterance
    length: stationary noise as timed (all-voiced), noise with quiet blocks (ragged), and the reference's end-to-end speech
    recording (testdata/librispeech_2.wav = tests/golden/e2e_0008.npz: 22.5 s whole, and its first two 10 s chunks). A mode is
    `tolerance_ok` only if the maximum over ALL of them is inside the tolerance."""
    import numpy as np
    from oracle import ktf_oracle as O
    whole, chunks = synth.speech_wavs(N)

    def windows(sec):          # windows of the speech recording at a 1 s hop (tests/test_gpu_margin.py)
        n = int(sec * 16000)
        return np.stack([whole[0, s0:s0 + n] for s0 in range(0, whole.shape[1] - n + 1, 16000)], 0)
    inputs = {"noise_as_timed": synth.make_wav(2, N, seed=1234), "noise_quiet_blocks": synth.make_wav(2, N, seed=4242, ragged=True),
              "speech_22s": whole, "speech_10s_chunks": chunks, "speech_5s_windows": windows(5.0), "speech_1.5s_windows": windows(1.5)}
    want = {k: O.xvector_forward(v, cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64) for k, v in inputs.items()}
    res = {"sample": f"fp64 NumPy oracle, 0008 topology, weights seed 4321: 2 + 2 synthetic utterances x {N} samples (all-voiced as timed / 30 % quiet "
                     f"0.5 s blocks) + the reference's e2e speech recording (359 665 samples whole, as two {N}-sample chunks, and as 5 s and "
                     f"1.5 s windows at a 1 s hop: 18 + 21 utterances). Each mode runs AS SHIPPED (Sequential.MIN_FRAMES and "
                     f"XvectorExtractor.route_short_utterances send utterances below 400 voiced frames of an f16mx model through the "
                     f"split-bf16 kernels); only the small-batch hand-over to the fp32 kernels (min_tiles) is off"}
    for g in modes:
        m = synth.build_extractor(ktf, cfg, w, gemm=g)
        m.xvec.min_tiles = {}          # a few utterances would be handed to the fp32 kernels: measure the mode's own (routing by length stays)
        r = {}
        for k, v in inputs.items():
            got = m(torch.as_tensor(v, device=dev)).cpu().numpy().reshape(v.shape[0], -1)
            r[k] = float(np.abs(got - want[k]).max())
        r["max"] = max(r.values())
        if g == "f16mx":               # for the record: the f16mx kernels themselves on the short windows (what the routing is for)
            m.xvec.min_frames, m.route_short_utterances = {}, False
            v = inputs["speech_1.5s_windows"]
            got = m(torch.as_tensor(v, device=dev)).cpu().numpy().reshape(v.shape[0], -1)
            r["unrouted_kernels_on_speech_1.5s_windows"] = float(np.abs(got - want["speech_1.5s_windows"]).max())
        res[g] = r
        del m
    torch.cuda.empty_cache()
    return res


def _timed_batch_vs_f32(torch, ktf, synth, cfg, w, wav, y):
    """Every x-vector of the TIMED batch against the exact fp32 kernels on the same waveforms (fp32 is 2e-6 from the fp64 oracle):
    the maximum, and high quantiles of the per-utterance maxima -- the tolerance claim on all B x 128 timed numbers, not on a sample."""
    m = synth.build_extractor(ktf, cfg, w, gemm="f32")
    ref = m(wav)
    per_utt = (y.double() - ref.double()).abs().amax(dim=1)
    q = torch.quantile(per_utt, torch.tensor([0.5, 0.99, 0.999], dtype=torch.float64, device=per_utt.device))
    del m
    torch.cuda.empty_cache()
    return {"utterances": int(per_utt.numel()), "max": float(per_utt.max()), "median": float(q[0]), "p99": float(q[1]), "p99.9": float(q[2]),
            "tolerance_ok": bool(float(per_utt.max()) <= TOLERANCE)}


def _roofline(gemm, gemm_stats, steps, B, T, passes):
    flops_per_step = B * T * FLOP_PER_FRAME_TDNN
    gemm_ms_per_step = gemm_stats["total_ms"] / steps
    achieved = flops_per_step / (gemm_ms_per_step * 1e-3) / 1e12
    peak = PEAK_TFLOPS[gemm]
    return {
        "bound": "mfma", "kernel": KERNELS[gemm],
        "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
        "mfma_passes_per_flop": passes,
        "mfma_issue_equivalent": achieved * passes,
        "frac_mfma_issue_equivalent": achieved * passes / peak,
        "launches_per_step": gemm_stats["launches"] // steps, "avg_launch_ms": gemm_stats["total_ms"] / max(gemm_stats["launches"], 1),
        "gemm_ms_per_step": gemm_ms_per_step, "per_layer_ms": gemm_stats["per_layer_ms"],
        "input_conversion_ms_per_step": gemm_stats.get("aux_ms_per_step", {}),      # ktf_mx_planes / ktf_split_bf16: part of the mode's cost, not of `achieved`
        "algorithmic_flop_per_step": flops_per_step,
        "per_layer_tflops": {k: _layer_flops(k, B, T) / (v * 1e-3) / 1e12 for k, v in gemm_stats["per_layer_ms"].items()},
        "note": ("achieved / frac count ALGORITHMIC flops (SURVEY 8d: 5 359 616 per voiced frame); this mode issues "
                 f"{passes:.3g} 16-bit-rate MFMA passes per algorithmic flop, so the matrix pipe is busy at frac_mfma_issue_equivalent")
                if passes > 1 else "",
    }


def _attach_traffic(roof, gemm, build_id):
    """HBM traffic of the GEMM launches comes from separate rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) committed
    under profiles/: it cannot be collected from inside this process. A stored figure belongs to the build it was measured on
    (`library_build_id` in the file = ktf_build_id() of that library): for any other build `traffic` stays null and the note says
    whose number is on file (the figure used to be pasted whatever the loaded library was)."""
    for tpath in (os.path.join(ROOT, "profiles", "r6", f"traffic_{gemm}.json"), os.path.join(ROOT, "profiles", "r5", f"traffic_{gemm}.json"),
                  os.path.join(ROOT, "profiles", "r4", f"traffic_{gemm}.json"), os.path.join(ROOT, "profiles", "r3", f"traffic_{gemm}.json")):
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            nl = max(roof["launches_per_step"], 1)
            alg = tj.get("tdnn_gemm_algorithmic_bytes_per_step")
            if tj.get("library_build_id") != build_id:
                roof["traffic_note"] = (f"no PMC traffic on file for this build ({build_id}); {os.path.relpath(tpath, ROOT)} holds "
                                        f"{tj['tdnn_gemm_bytes_per_step_corrected'] / nl:.4g} B per launch for build {tj.get('library_build_id', 'of an earlier round')}"
                                        + (f"; algorithmic {alg / nl:.4g} per launch" if alg else ""))
                return
            roof["traffic"] = tj["tdnn_gemm_bytes_per_step_corrected"] / nl
            roof["traffic_note"] = (f"HBM bytes per launch = PMC bytes per step ({os.path.relpath(tpath, ROOT)}, build {build_id}: "
                                    f"{tj['tdnn_gemm_bytes_per_step_corrected']:.4g}) / {nl} GEMM launches"
                                    + (f"; algorithmic {alg / nl:.4g} per launch" if alg else ""))
            return


def _scaling_summary(path, value, world):
    """Weak-scaling efficiency of this run against a stored N = 1 line of the same bench (per-GPU batch fixed)."""
    try:
        with open(path) as f:
            lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
        n1 = json.loads(lines[-1])
        if int(n1.get("n_gpus", 0)) != 1 or not n1.get("value"):
            return {"error": f"{path}: not an N = 1 line"}
        return {"n1_value": n1["value"], "n_gpus": world, "value": value, "efficiency_vs_n1": value / (world * n1["value"]),
                "n1_build_id": n1.get("library_build_id")}
    except (OSError, ValueError, IndexError) as e:
        return {"error": f"{path}: {e}"}


def _layer_flops(key, B, T):
    """FLOPs of all launches filed under a per_layer_ms key 'KxD->U[+stats]' (layers of equal shape share a key)."""
    kd, u = key.split("->")
    k, d = kd.split("x")
    n = {"3x512->512": 2}.get(key.replace("+stats", ""), 1)      # tdnn2 and tdnn3
    return 2.0 * B * T * int(k) * int(d) * int(u.replace("+stats", "")) * n


def _time_ms(torch, fn, iters, warm_s=0.15, min_s=0.05):
    """ms per call of fn in steady state: the GPU's clocks take ~0.1 s of load to settle after an idle gap (the first 20 steps of a
    256-utterance batch ran 7 % slower than the next 100: 2.63 against 2.45 ms), so the timed region starts after `warm_s` seconds of
    calls and lasts at least `iters` calls and `min_s` seconds."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        fn()
        torch.cuda.synchronize()         # (per call: a host that runs ahead would queue seconds of work in `warm_s` of wall time)
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(iters):
            fn()
        n += iters
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= min_s:
            return 1e3 * dt / n


def _other_configs(torch, ktf, synth, cfg, w, wav, gemm, dev, dev_info, mdl):
    """Side measurements of the remaining BASELINE.json configs on the same GPU (not part of `value`): the same
    1024-utterance step in the other GEMM arithmetic modes (each with its deviation from the fp64 oracle), a ragged
    variant of the workload, batch-1 latency (config 2), batch 256 (config 3) and the 1024 x 1024 PLDA trial matrix (config 5)."""
    import numpy as np
    from kaldi_tflite_amd import ops
    from oracle import ktf_oracle as O          # (the checker of the batch-1 x-vector below; never the thing measured)
    res = {}
    B = wav.shape[0]
    T = mdl.framing.numFrames(wav.shape[1])
    for g in ("f32", "bf16x3", "f16mx", "bf16"):
        if g == gemm:
            continue
        m = synth.build_extractor(ktf, cfg, w, gemm=g)
        ms = _time_ms(torch, lambda: m(wav), 5)
        res[g] = {"x_vectors_per_s": B / (ms * 1e-3), "ms_per_step": ms, "max_abs_dev_vs_fp64_oracle": dev_info[g]["max"],
                  "max_abs_dev_by_input": {k: v for k, v in dev_info[g].items() if k != "max"},
                  "tolerance_ok": bool(dev_info[g]["max"] <= TOLERANCE)}
        if g in ("bf16x3", "f16mx"):      # the modes that are compliant on any input: a driver-timed roofline block of their own
            prof = _GemmProfiler(ops, torch)
            for _ in range(3):
                m(wav)
            prof.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                m(wav)
            torch.cuda.synchronize()
            ms20 = 1e3 * (time.perf_counter() - t0) / 20
            res[g].update({"x_vectors_per_s": B / (ms20 * 1e-3), "ms_per_step": ms20, "steps": 20,
                           "roofline": _roofline(g, prof.finish(), 20, B, T, float(MFMA_PASSES[g]))})
            if B == 1024:
                _attach_traffic(res[g]["roofline"], g, ops.build_id())
        del m
        torch.cuda.empty_cache()
    # the same step with the shipped YAML's dither (data/tflite_models/0008_sitw_v2_1a.yml:43, `dither: 1.0`: on-device Philox + Box-Muller)
    md = synth.build_extractor(ktf, synth.extractor_cfg(dither=1.0), w, gemm=gemm)
    ms = _time_ms(torch, lambda: md(wav), 5)
    res[f"{gemm}_dither_1.0"] = {"x_vectors_per_s": B / (ms * 1e-3), "ms_per_step": ms, "note": "the reference's default front-end option; outputs are random by design"}
    del md
    # the same step on utterances with silence: 30 % of the 0.5 s blocks are quiet, the VAD drops them, batches are ragged
    quiet = (torch.rand((B, wav.shape[1] // 8000), device=dev, generator=torch.Generator(device=dev).manual_seed(5)) < 0.3)
    quiet[:, 0] = False
    gain = torch.where(quiet, 1e-3, 1.0).repeat_interleave(8000, dim=1)
    wav_r = torch.round(wav * gain)
    ms = _time_ms(torch, lambda: mdl(wav_r), 5)
    lens = mdl.last_lens.float()
    res[f"{gemm}_ragged"] = {"x_vectors_per_s": B / (ms * 1e-3), "ms_per_step": ms, "mean_voiced_frames": float(lens.mean()),
                             "min_voiced_frames": int(lens.min()), "note": "30 % of the 0.5 s blocks at 1e-3 gain: lens < T, compaction exercised"}
    del wav_r, gain, quiet
    # diarization-sized windows: 1024 x 1.5 s per step. The timed mode routes utterances under 400 frames to the split-bf16
    # kernels, which run such batches on flat row tiles (ktf_tdnn_split_flat)
    short = torch.clamp(torch.round(1000.0 * torch.randn((B, 24000), generator=torch.Generator(device=dev).manual_seed(77), device=dev)), -32767, 32767)
    ms = _time_ms(torch, lambda: mdl(short), 5)
    res[f"{gemm}_1.5s_windows"] = {"x_vectors_per_s": B / (ms * 1e-3), "ms_per_step": ms, "note": "1024 windows of 1.5 s (148 frames) per step"}
    if gemm != "f32":
        # ... with the roofline of the kernels the route runs: for the timed mode the split-bf16 plane kernels on flat row tiles (csrc/tdnn_split.hip,
        # tdnn_x3s_kernel<flat> and <flat, pooled>), three bf16 MFMA passes per algorithmic flop, against the dense 16-bit peak
        route = mdl.xvec.SHORT_MODE.get(gemm, gemm) if mdl.xvec.frames_floor(gemm) > 148 else gemm
        prof = _GemmProfiler(ops, torch)
        for _ in range(3):
            mdl(short)
        prof.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            mdl(short)
        torch.cuda.synchronize()
        ms20 = 1e3 * (time.perf_counter() - t0) / 20
        Ts = mdl.framing.numFrames(short.shape[1])
        roof = _roofline(route, prof.finish(), 20, B, Ts, float(MFMA_PASSES[route]))
        if route == "bf16x3":
            roof["kernel"] = "tdnn_x3s_kernel<flat> (tdnn1-4) + tdnn_x3s_kernel<flat, pooled> (tdnn5 + pooling), csrc/tdnn_split.hip"
        dv = dev_info.get(gemm, {}).get("speech_1.5s_windows")
        res[f"{gemm}_1.5s_windows"].update({"ms_per_step_with_per_launch_events": ms20, "route": route, "roofline": roof,
                                            "max_abs_dev_vs_fp64_oracle_on_speech_1.5s_windows": dv})
    del short
    # int16 PCM input: same step, half the input bytes; and the PCIe-inclusive rate of a host-fed step
    wav16 = wav.to(torch.int16)
    res[f"{gemm}_int16_input"] = {"x_vectors_per_s": B / (_time_ms(torch, lambda: mdl(wav16), 5) * 1e-3)}
    host16 = wav16.cpu().pin_memory()
    res[f"{gemm}_int16_from_pinned_host"] = {"x_vectors_per_s": B / (_time_ms(torch, lambda: mdl(host16.to(dev, non_blocking=True)), 5) * 1e-3),
                                             "note": "PCIe-inclusive, upload serialised in front of each step"}
    hb = [host16] * 8

    def streamed():
        for _ in mdl.extract_stream(hb, depth=3):
            pass
    res[f"{gemm}_int16_from_pinned_host_overlapped"] = {"x_vectors_per_s": 8 * B / (_time_ms(torch, streamed, 3) * 1e-3),
                                                       "note": "PCIe-inclusive, 8 batches, uploads on a second stream (first upload exposed)"}
    del wav16, host16, hb
    torch.cuda.empty_cache()
    # BASELINE config 3: batch 256, bf16 TDNN
    m256 = synth.build_extractor(ktf, cfg, w, gemm="bf16")
    x256 = wav[:256].contiguous()
    ms = _time_ms(torch, lambda: m256(x256), 10)
    res["config3_batch256_bf16"] = {"x_vectors_per_s": 256 / (ms * 1e-3), "ms_per_step": ms, "max_abs_dev_vs_fp64_oracle": dev_info["bf16"]["max"],
                                    "tolerance_ok": bool(dev_info["bf16"]["max"] <= TOLERANCE)}
    del m256
    m256 = synth.build_extractor(ktf, cfg, w, gemm="f16mx")       # the same batch in the fastest mode inside the tolerance
    ms = _time_ms(torch, lambda: m256(x256), 10)
    res["config3_batch256_f16mx"] = {"x_vectors_per_s": 256 / (ms * 1e-3), "ms_per_step": ms, "max_abs_dev_vs_fp64_oracle": dev_info["f16mx"]["max"],
                                     "tolerance_ok": bool(dev_info["f16mx"]["max"] <= TOLERANCE)}
    del m256
    # BASELINE config 2: batch 1 — eager launches and the captured hipGraph (XvectorExtractor.compile), per mode: "f32" (the exact
    # fp32 small-tile kernels; batch == single bitwise) and the timed mode (a batch this small runs the bf16-pair small tiles,
    # KTF_GEMM_BF16X4: Sequential.small_tile_pairs)
    one = wav[:1].contiguous()
    want1 = O.xvector_forward(one.cpu().numpy(), cfg, synth.oracle_layers(w), w["mean"], w["lda"], dtype=np.float64)
    res["config2_batch1_latency_ms"] = {}
    for mode in dict.fromkeys(("f32", gemm)):
        m1 = synth.build_extractor(ktf, cfg, w, gemm=mode)
        r = {"eager": _time_ms(torch, lambda: m1(one), 50)}
        run = m1.compile(one)
        r["hipgraph"] = _time_ms(torch, lambda: run(one), 50)
        r["hipgraph_bitwise_equal_eager"] = bool(torch.equal(run(one), m1(one)))
        r["max_abs_dev_vs_fp64_oracle"] = float(np.abs(m1(one).cpu().numpy().reshape(1, -1) - want1).max())
        res["config2_batch1_latency_ms"][mode] = r
        del m1, run
    # BASELINE config 5: 1024 x 1024 PLDA trial matrix on the x-vectors of this batch
    rng = np.random.default_rng(31)
    dim, nb = 128, 1024
    A = rng.standard_normal((dim, dim)) / np.sqrt(dim) + np.eye(dim)
    plda = ktf.layers.PLDA(dim, rng.standard_normal(dim) * 0.1, A, np.sort(rng.uniform(0.05, 30.0, dim))[::-1].copy())
    xv = mdl(wav)[:nb].to(torch.float64)
    res["config5_plda_1024x1024_fp64_ms"] = _time_ms(torch, lambda: plda(xv), 10)
    return res


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cpus():
    """Logical CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_baseline(torch, synth, cfg, w, N):
    """SURVEY §8(d) / BASELINE.md §3: the torch-CPU fp32 restatement of the reference's op graph (materialised frames,
    rfft, dense mel matmul, materialised im2col + matmul; constants precomputed once) on the host cores: 3 warm-ups, >= 10
    timed iterations, median; the full extractor one utterance at a time (the reference's own batch size) with intra-op
    threads, and utterance-parallel (one single-threaded extraction per usable core), the Framing + MFCC + CMVN configuration
    (BASELINE config 1) and MFCC alone. The thread count is calibrated first (a short
    B = 8 run per candidate, ascending, stop when it gets slower): with every logical CPU of a shared 256-thread host in the
    pool, torch's intra-op barriers cost more than the extra cores give (measured here: 7.9 s per utterance at 256 threads
    against tens of milliseconds at 16-64). `cores` = the threads actually used; `value` = the better full-extractor rate."""
    import numpy as np
    from oracle.ktf_torch_cpu import KtfRef
    avail = _usable_cpus()
    ref = KtfRef(cfg, synth.oracle_layers(w), w["mean"], w["lda"])
    cal = torch.as_tensor(synth.make_wav(8, N, seed=99))
    tried, best_t, best_s = {}, None, None
    for t in [c for c in (8, 16, 32, 64, 128, 256, 512) if c < avail] + [avail]:
        torch.set_num_threads(t)
        ref(cal)
        t0 = time.perf_counter()
        ref(cal)
        s = time.perf_counter() - t0
        tried[t] = s
        if best_s is None or s < best_s:
            best_t, best_s = t, s
        elif s > 1.5 * best_s:
            break
    torch.set_num_threads(best_t)

    def median_s(fn, warm=3, iters=10, budget_s=10.0):
        for _ in range(warm):
            fn()
        ts, t_all = [], time.perf_counter()
        while len(ts) < iters or (time.perf_counter() - t_all < 1.0 and len(ts) < 200):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget_s and len(ts) >= iters:
                break
        return float(np.median(ts)), len(ts)

    T = 1 + (N - 400) // 160
    legs = {}
    wav = torch.as_tensor(synth.make_wav(1, N, seed=1234))
    s, n = median_s(lambda: ref(wav))
    legs["extractor_B1"] = {"x_vectors_per_s": 1 / s, "median_s": s, "iterations": n, "threads": best_t,
                            "note": "one utterance at a time (the reference's own batch size), intra-op threads"}
    # utterance-parallel: one single-threaded extraction per core (the per-utterance working set stays in the core's caches; the
    # batched B = 32 call of rounds 1-2 materialised 196 MB of im2col per layer and ran 3.7 x slower per utterance than B = 1)
    import concurrent.futures
    torch.set_num_threads(1)
    wavs = [torch.as_tensor(synth.make_wav(1, N, seed=2000 + i)) for i in range(2 * avail)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=avail) as pool:
        list(pool.map(ref, wavs[:avail]))                 # warm-up
        rates = []
        for _ in range(3):
            t0 = time.perf_counter()
            list(pool.map(ref, wavs))
            rates.append(len(wavs) / (time.perf_counter() - t0))
    legs["extractor_utterance_parallel"] = {"x_vectors_per_s": float(np.median(rates)), "workers": avail, "utterances_per_round": len(wavs),
                                            "rounds": 3, "note": "one intra-op thread per worker, one utterance per worker at a time"}
    torch.set_num_threads(best_t)
    wav1 = torch.as_tensor(synth.make_wav(1, N, seed=1234))
    s, n = median_s(lambda: ref.features(wav1))
    legs["framing_mfcc_cmvn_B1"] = {"frames_per_s": T / s, "median_s": s, "iterations": n}
    wav32 = torch.as_tensor(synth.make_wav(32, N, seed=1234))
    s, n = median_s(lambda: ref.mfcc(wav32))
    legs["mfcc_only_B32"] = {"frames_per_s": 32 * T / s, "median_s": s, "iterations": n}
    par = legs["extractor_utterance_parallel"]["x_vectors_per_s"] > legs["extractor_B1"]["x_vectors_per_s"]
    best = max(legs["extractor_B1"]["x_vectors_per_s"], legs["extractor_utterance_parallel"]["x_vectors_per_s"])
    return {"value": best, "unit": "x-vectors/s", "cores": avail if par else best_t, "kind": "port", "cpu_model": _cpu_model(),
            "host_logical_cpus": os.cpu_count(), "usable_cpus": avail,
            "thread_calibration_s_per_8_utterances": {str(k): round(v, 4) for k, v in tried.items()},
            "sample": (f"CPU restatement ({best_t} threads of {avail} usable CPUs): torch-CPU fp32 port of the reference's TF "
                       f"op graph (oracle/ktf_torch_cpu.py; the reference's TensorFlow 2.8 cannot be installed here), {N}-sample "
                       f"utterances of the same workload, 3 warm-ups + >= 10 timed iterations per leg, median; value = best "
                       f"of one utterance at a time with {best_t} intra-op threads / {avail} single-threaded extractions in parallel"),
            "legs": legs}


if __name__ == "__main__":
    main()
