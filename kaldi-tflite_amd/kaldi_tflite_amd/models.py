"""
ktf.models — SequentialFromConfig / XvectorExtractor / XvectorExtractorFromConfig with the
reference's signatures (kaldi_tflite/lib/models/kaldi/{sequential,xvector_extractor}.py).

The builders read the same two-level YAML as the reference. At call time the model does not
run layer by layer: the canonical chains are recognised and dispatched to fused kernels
(Framing+MFCC in one launch, VAD+compaction+CMVN in one launch, every [affine, relu, batchnorm]
triple as one MFMA GEMM with a fused epilogue), on ragged utterance-strided activations so a
batch of B utterances equals B independent batch-1 calls of the reference.
"""

import math
import os
import threading

import numpy as np
import torch

from . import _lib as L
from . import ops
from .io import KaldiNnet3Reader, ReadKaldiArray
from .layers import TDNN, BatchNorm, CMVN, Framing, MFCC, ReLU, StatsPooling, VAD, _GEMM, WEIGHTS_EPOCH
from .mx import Planes


class Input:
    """Placeholder for keras.layers.Input: carries the declared (batch, time, feat) shape."""

    def __init__(self, shape=None, batch_size=None, name="input"):
        self.shape = (batch_size,) + tuple(shape)
        self.name = name


# ------------------------------------------------------------------------------------------------ config -> layers
# One row per layer kind of the model YAML (models/kaldi/sequential.py:29-83): the spellings the reference accepts, the
# layer class, how the layer is named from the entry's `name`, and whether the entry's `cfg` dict is its kwargs.
_LAYER_KINDS = (
    (("affine", "tdnn"), TDNN, "{}.affine", True),
    (("relu",), ReLU, "{}.relu", False),
    (("batchnorm", "bn"), BatchNorm, "{}.batchnorm", False),
    (("stats", "stats_extraction", "stats_pooling"), StatsPooling, "{}", True),
)
_KIND_OF = {alias: row for row in _LAYER_KINDS for alias in row[0]}


def cfg2layers(layerCfg):
    """One entry of `model_config.layers` -> the layers it stands for, in order. `type` is a kind or a list of kinds
    ("affine" | "tdnn", "relu", "batchnorm" | "bn", "stats" | "stats_extraction" | "stats_pooling"); the entry's `cfg`
    holds the keyword arguments of its parametrised layer. KeyError without a `type`, ValueError for an unknown one."""
    kinds = layerCfg.get("type") or []
    if isinstance(kinds, str):
        kinds = [kinds]
    if not kinds:
        raise KeyError("layer config does not define layer 'type'")
    built = []
    for kind in kinds:
        row = _KIND_OF.get(kind.lower())
        if row is None:
            raise ValueError(f"unsupported layer type '{kind.lower()}'")
        _, cls, name_fmt, takes_cfg = row
        kwargs = dict(layerCfg.get("cfg") or {}) if takes_cfg else {}
        kwargs["name"] = name_fmt.format(layerCfg.get("name"))
        built.append(cls(**kwargs))
    return built


class _Workspace:
    """Device scratch of one model. One byte arena per (role, device, stream, host thread), grown to the largest request seen
    and re-viewed for every call: memory is bounded by the largest batch processed, not by the number of distinct
    (batch, length) shapes (variable-length audio used to allocate a full activation set per shape). Arenas are keyed by the
    CURRENT stream and by the CALLING THREAD: two streams never share scratch, and two host threads driving ONE model never do
    either, not even on one stream (threads that set no stream all sit on the default one, where their launches interleave:
    each call's kernels must then find their own buffers). Calls of one thread on one stream are ordered by the stream.
    The binding made by enter() is per thread as well."""

    def __init__(self):
        self._arenas = {}
        self._tl = threading.local()

    @property
    def _where(self):
        return getattr(self._tl, "where", None)

    def enter(self, device):
        """Bind the following get() calls of this thread to `device` and its current stream (looked up once per model call)."""
        self._tl.where = (str(device), torch.cuda.current_stream(device).cuda_stream, threading.get_ident())

    def get(self, role, shape, dtype, device, padded=True):
        """`padded`: the view has pad columns nobody writes (they must read as finite zeros): True zeroes the whole view when
        the role's shape changes, an int zeroes the columns from that index on only; False hands the re-sliced bytes out as
        they are (every kernel clamps its row reads to lens[b] - 1 and skips utterances with lens[b] == 0, so rows nobody
        wrote are never read)."""
        if self._where is None or self._where[0] != str(device):
            self.enter(device)
        key = (role,) + self._where
        sig = (tuple(shape), dtype)
        slot = self._arenas.get(key)             # [arena bytes, signature of the current view, the view]
        if slot is not None and slot[1] == sig:
            return slot[2]
        count = int(math.prod(shape))
        nbytes = max(count * dtype.itemsize, 16)
        if slot is None or slot[0].numel() < nbytes:
            slot = [torch.zeros((nbytes,), dtype=torch.uint8, device=device), None, None]
            self._arenas[key] = slot
        view = slot[0][:nbytes].view(dtype)[:count].view(shape)
        if slot[1] is not None and padded is not False:
            # a different shape re-slices old bytes: pad columns must read as finite zeros
            (view if padded is True else view[..., int(padded):]).zero_()
        slot[1], slot[2] = sig, view
        return view

    def bytes(self):
        return sum(s[0].numel() for s in self._arenas.values())

    def clear(self):
        self._arenas.clear()


class DeferredTail:
    """run_ragged(defer_tail=True): the stack up to the pooling has run; what is left -- (finalize of the pooled sums,) the affine
    after the pooling -- is handed to the caller, who fuses it with its own post-processing (XvectorExtractor: ktf_xvec_tail_f32)."""

    def __init__(self, layer, B, D, include_std, eps, pooled=None, sums=None, slots=0, lens=None, T=0, slot_rows=128):
        self.layer, self.B, self.D, self.include_std, self.eps = layer, B, D, include_std, eps
        self.pooled, self.sums, self.slots, self.lens, self.T, self.slot_rows = pooled, sums, slots, lens, T, slot_rows


class Sequential:
    """Keras-Sequential stand-in: `.layers`, `.name`, `mdl(x, training=False)`, `get_layer`, `summary`.
    `gemm` selects the TDNN arithmetic of the fused runner: "f32" (exact fp32 MFMA, default = the reference's
    precision), "bf16x3" (split-bf16, fp32-grade), "f16mx" (one half-precision pass + two block-scaled residual passes:
    inside the 1e-4 x-vector tolerance at half the split-bf16 matrix work, the timed mode), "bf16" (one pass, outside that tolerance)."""

    split_planes = True         # bf16x3: hi/lo bf16 activation planes between the wide layers (no in-loop conversion)
    # (the split-plane layers always walk the contexts of a multi-context layer inside each 32-feature chunk -- KTF_TDNN_K_INTERLEAVED: L2 reuse --
    #  and read their weights as the kernel's LDS stage images -- KTF_TDNN_W_TILED: 1 KiB per DMA instruction; both were A/B switches until round 6)
    # Routing defaults, copied into every instance (`self.min_tiles`, `self.min_frames`: per-model knobs, no shared mutable state).
    # MIN_TILES: batches with fewer 256-row tiles than this run on the exact fp32 kernels (crossover of the measured per-layer times).
    # MIN_FRAMES: batches whose utterances are shorter than this many frames run the next tighter mode: the block-scaled residuals
    # of f16mx are zero-mean rounding noise that the statistics pooling averages over the voiced frames, so its deviation grows as
    # the utterance shrinks (measured on speech windows with BatchNorm statistics that are the network's own, tests/test_gpu_margin.py:
    # 10 s 2-5e-5, 5 s 5-6.5e-5, 3 s 6-7.5e-5, 1.5 s up to 1.2e-4 -- outside the 1e-4 tolerance); below 4 s the split-bf16 kernels
    # (1.5e-5) take the batch, and XvectorExtractor applies the same rule per utterance on the device (route_short_utterances).
    MIN_TILES = {"bf16": 12, "bf16x3": 32, "f16mx": 20}      # (tools/small_batch_crossover.py, 10 s utterances =
                                                                                     # 4 tiles each: bf16 from 3, f16mx from 5, bf16x3 from 8)
    MIN_FRAMES = {"f16mx": 400}
    SHORT_MODE = {"f16mx": "bf16x3"}
    # ... and where the weights themselves are what the block-scaled images like least: one E8M0 scale per 32 weights means that a weight
    # tens of standard deviations out takes the scale of its block with it and the other 31 are rounded against it. Measured
    # (tests/test_gpu_margin.py, round 5: Student-t rows with outliers of 30-150 sigma): one of 128 windows of 500-540 frames at 8.9e-5
    # where Gaussian rows and BatchNorm variances over two decades stay below 5.5e-5, and 2.6e-5 at 10 s. A model whose frame-level
    # layers hold a weight beyond OUTLIER_SIGMAS standard deviations of its layer routes below MIN_FRAMES_OUTLIERS instead
    # (frames_floor; trained x-vector networks: unknown until the pretrained weights are at hand -- verify_fraction measures them).
    OUTLIER_SIGMAS = 20.0
    MIN_FRAMES_OUTLIERS = {"f16mx": 640}

    def __init__(self, layers=None, name=None, gemm="f32"):
        self.input = None
        self.layers = []
        for l in layers or []:
            if isinstance(l, Input):
                self.input = l
            else:
                self.layers.append(l)
        self.name = name if name is not None else "sequential"
        if gemm not in _GEMM:
            raise ValueError(f"gemm must be one of {sorted(_GEMM)}")
        self.gemm = gemm
        self.min_tiles = dict(self.MIN_TILES)
        self.min_frames = dict(self.MIN_FRAMES)
        self.min_frames_outliers = dict(self.MIN_FRAMES_OUTLIERS)
        self._outlier_score = (None, 0.0)                 # (weights signature, largest |w| / std(w) over the frame-level layers)
        self.mx_loader = None        # f16mx: which kernel runs the frame-level layers. False = the 256 x 256 eight-wave kernel (csrc/tdnn_mx.hip),
                                     # True = the loader-wave kernel (csrc/tdnn_mxl.hip: 192 x 256 tiles on the flat row space, eight matrix +
                                     # four loader waves), None = per call, whichever needs less time by rounds of 256 workgroups x rows
                                     # per tile (_mx_use_loader): the 256-row kernel is 4-6 % faster per unit of work
                                     # and takes every batch that fills the chip; 5 ... 24 and ~48 utterances of 10 s leave it
                                     # partly empty and run 8-16 % faster on the smaller flat tiles (tools/mid_batch.py)
        self.mx_flat_rows = True     # f16mx, 256-row kernel, plane outputs: M-tiles over the batch's valid rows laid end to end where that saves
                                     # tiles (ktf_tdnn_mx_flat: 998-frame utterances fill 3.9 of their 4 tiles; bit-identical planes)
        self.flat_rows_long = True   # ... and of any batch whose utterances do not fill their last tile (like mx_flat_rows)
        self.flat_rows = True        # bf16x3 plane layers of short utterances: M-tiles over the valid rows laid end to end (ktf_tdnn_split_flat)
        self.flat_pooling = True     # ... the layer pooled in its epilogue included (ktf_tdnn_split_flat_stats, ktf_tdnn_mx_flat_stats). Its partial sums are cut along the
                                     # flat row space: reproducible run to run, but an utterance's pooled values can differ in the last bits of the
                                     # fp32 partial sums (<= 2e-6 on an x-vector) with the batch it arrives in; False: per-utterance tiles for
                                     # that layer, x-vectors of the split-bf16 route independent of the batch bit for bit, 3.5 % slower on 1.5 s windows
        self.small_tile_pairs = True # batches below `min_tiles` of a reduced-precision model: bf16-pair small tiles instead of fp32 ones
        self.fuse_stats = True       # pool inside the epilogue of the GEMM that feeds a reducing StatsPooling
        self.deterministic = True    # ... with per-block partial sums added in a fixed order (bitwise reproducible runs)
        self.dtype = "float32"
        self._ws_own = _Workspace()
        self._tl = threading.local()  # per host thread: a workspace override (XvectorExtractor.compile), the deferred tail of the call in flight
        self._build()

    def _build(self):
        """Build every layer from the declared input feature dim so weights can be imported before the first call."""
        if self.input is None or self.input.shape[-1] is None:
            return
        shape = (self.input.shape[0], self.input.shape[1], self.input.shape[-1])
        for l in self.layers:
            if not l.built:
                l.build(shape)
            shape = tuple(l.compute_output_shape(shape))

    def weight_outlier_score(self):
        """Largest |w| / std(w) over the wide frame-level TDNN layers (cached per weights signature)."""
        sig = self.weights_signature()
        if self._outlier_score[0] != sig:
            score = 0.0
            for l in self.layers:
                if isinstance(l, TDNN) and l.built and l.units > 128:
                    k = np.asarray(l.kernel, np.float64)
                    sd = float(k.std())
                    if sd > 0:
                        score = max(score, float(np.abs(k).max()) / sd)
            self._outlier_score = (sig, score)
        return self._outlier_score[1]

    def frames_floor(self, mode):
        """Utterances (batches) of fewer frames than this run `SHORT_MODE[mode]` instead of `mode`: `min_frames`, or `min_frames_outliers`
        for a model with a weight beyond OUTLIER_SIGMAS standard deviations of its layer (0: no floor)."""
        base = self.min_frames.get(mode, 0)
        if base and mode in self.min_frames_outliers and self.weight_outlier_score() > self.OUTLIER_SIGMAS:
            return max(base, self.min_frames_outliers[mode])
        return base

    def weights_signature(self):
        """Changes whenever a TDNN / BatchNorm layer's weights change (set_weights, a re-build): captured graphs
        are tied to it."""
        return tuple((id(l), l._version) for l in self.layers if isinstance(l, (TDNN, BatchNorm)))

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError(f"No such layer: {name}.")

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        for l in self.layers:
            print_fn(f"  {l.name:32s} {type(l).__name__}")

    # ------------------------------------------------------------------ fused execution plan
    def _plan(self):
        """Group [TDNN(no act), ReLU, BatchNorm] runs; returns None if a layer outside the fusable set is present."""
        steps, i, Ls = [], 0, self.layers
        pooled = False
        while i < len(Ls):
            l = Ls[i]
            if isinstance(l, TDNN):
                # behind a reducing StatsPooling the runner stacks the pooled vectors as ONE B-row matrix: right for a layer that looks
                # at its own frame only (every x-vector network); any other context would read the neighbouring UTTERANCES' rows
                if pooled and (list(l.context) != [0] or l.subsamplingFactor != 1):
                    return None
                relu, bn = False, None
                j = i + 1
                if l.activation in (None, "linear") and j < len(Ls) and isinstance(Ls[j], ReLU):
                    relu = True
                    j += 1
                if (relu or l.activation in (None, "linear", "relu")) and j < len(Ls) and isinstance(Ls[j], BatchNorm):
                    bn = Ls[j]
                    j += 1
                steps.append(("tdnn", l, relu, bn))
                i = j
            elif isinstance(l, StatsPooling) and l.reduce:
                steps.append(("stats", l))
                pooled = True
                i += 1
            elif isinstance(l, (ReLU, BatchNorm)):
                steps.append(("eltwise", l))
                i += 1
            else:
                return None
        return steps

    def batch_gemm(self, B, T, mode=None):
        """GEMM arithmetic for a batch of B utterances of up to T frames: the model's mode, except that a handful of
        256-row tiles (single utterances) cannot fill the chip on the 256-wide ring kernels -- the exact fp32 kernels have
        small-tile forms and are faster there (one 10 s utterance: 0.19 ms against 0.20 bf16 / 0.39 split-bf16); and a batch
        of utterances shorter than `min_frames` frames runs the tighter mode named by SHORT_MODE."""
        return self._batch_route(B, T, mode)[0]

    def _batch_route(self, B, T, mode=None):
        """-> (batch_gemm, pairs). `pairs`: a batch of a reduced-precision model that is too small for the 256-row tiles runs its wide
        layers on the bf16-pair small tiles (KTF_GEMM_BF16X4, csrc/tdnn_pair.hip) instead of the fp32 ones: the first layer (fp32
        kernel) writes pairs, the layers behind it read and write pairs, the last one in front of the pooling writes fp32."""
        if mode is None:
            mode = self.gemm
            if T < self.frames_floor(mode):              # short utterances: the tighter mode (MIN_FRAMES)
                mode = self.SHORT_MODE.get(mode, "f32")
        gemm = _GEMM[mode]
        # (the batch's rows in units of 256: what the small tiles' time follows -- a 3 s utterance is 1.2 of them, not two tiles;
        # measured crossovers at 1.5 / 3 / 10 s: 28-32 / 24-28 / 7 utterances for "bf16x3", tools/small_batch_crossover.py)
        if gemm != L.GEMM_F32 and -(-(B * T) // 256) < self.min_tiles.get(mode, 0):
            return L.GEMM_F32, bool(self.small_tile_pairs)
        return gemm, False

    # ---- per-thread call state: a model object may be driven by several host threads at once (the reference's layers are stateless
    # after build: layers/tdnn/tdnn.py:251-280, normalization/cmvn.py:186-250), so nothing a call sets lives on the instance
    @property
    def _ws(self):
        return getattr(self._tl, "ws", None) or self._ws_own

    @_ws.setter
    def _ws(self, ws):                       # XvectorExtractor.compile: private workspaces for the capturing thread only
        self._tl.ws = None if ws is self._ws_own else ws

    @property
    def _deferred(self):
        return getattr(self._tl, "deferred", None)

    @_deferred.setter
    def _deferred(self, d):
        self._tl.deferred = d

    def _mx_use_loader(self, B, T):
        """f16mx: the loader-wave kernel for this batch? (`mx_loader` None: rounds of 256 workgroups x rows per tile, two N-tiles)"""
        if self.mx_loader is not None:
            return bool(self.mx_loader)
        rounds = lambda wgs: -(-wgs // 256)  # noqa: E731
        cost_256 = rounds(2 * B * ((T + 255) // 256)) * 256
        cost_loader = rounds(2 * ((B * T + 191) // 192)) * 192 * 1.05
        return cost_loader < cost_256

    def _flat_tiles(self, l, B, T, ldx):
        """Split-bf16 plane layers of utterances that fill their 256-row tiles badly (a 1.5 s window: 148 rows) run on M-tiles over the
        batch's valid rows laid end to end (ktf_tdnn_split_flat*)."""
        if not (self.flat_rows and B * T > 0 and l.padding == "SAME" and l.subsamplingFactor == 1 and l.activation in (None, "linear", "relu")
                and B <= 4095 and B * T * ldx * 2 < 2 ** 32):
            return False
        # tiles that are mostly padding (fewer than 80 % of the 16-row blocks computed hold a row of a full-length utterance), or at least
        # 1.5 % fewer tiles even if the VAD dropped no frame (a ragged batch saves half a tile per utterance on top)
        return bool(5 * (-(-T // 16)) < 4 * (-(-T // 256)) * 16 or (self.flat_rows_long and -(-(B * T) // 256) * 200 <= B * (-(-T // 256)) * 197))

    def _pooled_by_gemm(self, l, relu, bn, nxt, x_or_planes, lens, gemm, split, dev, T, defer_to=None, row_starts=None):
        """[affine, relu, batchnorm] -> reducing StatsPooling inside the GEMM epilogue: the layer output is never written.
        `split`: the input is a (2,B,T,ld) pair of bf16 planes read by the split-plane kernel; `row_starts`: ... on flat row tiles
        (reproducible form only). Returns the pooled (1, B, od) view."""
        sp = nxt[1]
        D = l.units
        od = 2 * D if sp.includeStd else D
        ld = ops.round_up(od, 32)
        B = x_or_planes.shape[1] if x_or_planes.dim() == 4 else x_or_planes.shape[0]
        flat = row_starts is not None
        slots = (ops.flat_stats_slots(T) if flat else ops.stats_slots(T)) if self.deterministic else 0
        sums = self._ws.get("sums", (B, max(slots, 1), 2, D), torch.float64, dev, padded=False)
        sbuf = self._ws.get("pooled", (B, ld), torch.float32, dev, padded=od if ld != od else False)
        kint = bool(split and l.kernelWidth > 1)
        wt = bool(split)
        w, w_lo, bias = l.device_weights(dev, gemm, k_interleaved=kint, w_tiled=wt)
        scale, shift = bn.affine_device(dev) if bn is not None else (None, None)
        xdt = x_or_planes.dtype
        d = l.desc(gemm, xdt, xdt if split else L.act_torch_dtype(gemm), act="relu" if relu else None,
                   flags=(L.TDNN_DET_STATS if slots else 0) | (L.TDNN_K_INTERLEAVED if kint else 0) | (L.TDNN_W_TILED if wt else 0))
        if flat:
            ops.tdnn_split_flat_stats(x_or_planes, row_starts, d, w, w_lo, bias, scale, shift, sums, zero=not slots)
            if slots:
                ops.stats_finalize_flat(sums, row_starts, T, D, sp.includeStd, sp.epsilon, sbuf, slots)
            else:
                ops.stats_finalize(sums, lens, T, D, sp.includeStd, sp.epsilon, sbuf)
            if defer_to is not None:     # (the fused tail reads finished pooled rows here: its own finalize knows per-utterance slots only)
                self._deferred = DeferredTail(defer_to, B, D, sp.includeStd, sp.epsilon, pooled=sbuf, lens=lens, T=T)
            return sbuf[:, :od].unsqueeze(0)
        (ops.tdnn_split_stats if split else ops.tdnn_stats)(x_or_planes, lens, d, w, w_lo, bias, scale, shift, sums, zero=not slots)
        if defer_to is not None:         # the caller's fused tail finalizes the sums itself
            self._deferred = DeferredTail(defer_to, B, D, sp.includeStd, sp.epsilon, sums=sums, slots=slots, lens=lens, T=T)
            return sbuf[:, :od].unsqueeze(0)
        ops.stats_finalize(sums, lens, T, D, sp.includeStd, sp.epsilon, sbuf, slots=slots)
        return sbuf[:, :od].unsqueeze(0)

    def _tail_step(self, steps):
        """Index of the last step if it is a plain affine (context [0], no activation / BatchNorm) right after a reducing
        StatsPooling -- the tail XvectorExtractor fuses with its LDA / length-norm step -- else -1."""
        if len(steps) >= 2 and steps[-1][0] == "tdnn" and steps[-2][0] == "stats":
            _, l, relu, bn = steps[-1]
            if (not relu and bn is None and l.activation in (None, "linear") and l.kernelWidth == 1 and l.padding == "SAME"
                    and l.subsamplingFactor == 1 and l.units <= 512 and l.inputDim <= 3072):
                return len(steps) - 1
        return -1

    def run_ragged(self, x, lens=None, defer_tail=False, mode=None):
        """x: (B, T, D) view of an utterance-strided buffer whose row stride is a multiple of 8 and >= round_up(D, 32)
        (pad columns finite); lens: int32 (B,) valid rows per utterance or None. Returns (B, T', units) for frame-level
        outputs or (B, 1, units) after a reducing StatsPooling. The result is a view of this model's workspace: it is
        overwritten by the model's next call on the same stream (`__call__` hands out an owned copy). `mode`: run this call in
        that arithmetic instead of the model's (XvectorExtractor's second pass over short utterances)."""
        steps = self._plan()
        if steps is None:
            raise NotImplementedError("this layer stack is not supported by the fused ragged runner")
        dev = x.device
        self._ws.enter(dev)
        gemm, pairs = self._batch_route(x.shape[0], x.shape[1], mode=mode)
        x_pair = False                                   # x holds KTF_BF16P pairs (in a float32 tensor)
        rows_T = [None]                                  # the frame count the flat-row bookkeeping below was made for

        def flat_rows_for(rs, B_, T_):
            """The flat-row bookkeeping of the current (lens, B, T): made when the first flat layer runs, made again when a VALID-padded or
            subsampling layer has changed the lengths (the callers reset it) OR the frame count -- a dense batch (lens None) has no lengths to
            replace, only T changes (tools/fuzz_models.py seeds 202 / 203: stale rows behind a VALID-padded layer of a dense batch)."""
            if rs is None or rows_T[0] != (B_, T_):
                rs = ops.flat_rows(lens, B_, T_, lambda role, shape, dt: self._ws.get(role, shape, dt, dev, padded=False))
                rows_T[0] = (B_, T_)
            return rs
        row_starts = None                                # prefix sums of lens for the layers on flat row tiles: made when the first one runs,
                                                         # again when a VALID-padded or subsampling layer has changed the lengths
        act_dtype = L.act_torch_dtype(gemm)
        tail_at = self._tail_step(steps) if defer_tail else -1
        self._deferred = None        # set by a pooling step whose consumer is the deferred tail
        pooled = False
        skip = False
        # split-bf16 mode: frame-level activations travel between the wide layers as hi/lo bf16 planes (2,B,T,ld) instead
        # of fp32, so the GEMM K-loop carries no conversion (ktf_tdnn_split); `planes` holds them while they exist
        use_planes = gemm == L.GEMM_BF16X3 and self.split_planes
        planes = None
        pending_bn = None            # F16MX: the BatchNorm of the previous layer, to be folded into the next layer's weights
        mxp = None                   # F16MX: the current activations as the four MX planes (mx.Planes)
        for si, st in enumerate(steps):
            if skip:
                skip = False
                continue
            nxt = steps[si + 1] if si + 1 < len(steps) else None
            # layer si reads what layer si-1 wrote: two arenas alternate; outputs with pad columns and the single rows
            # after the pooling get arenas of their own, so that in steady state no role changes shape (a change costs a fill)
            out_role = f"act{si & 1}" + ("s" if pooled else "")
            if st[0] == "tdnn" and st[1].units % 32:
                out_role += "p"
            if si == tail_at and self._deferred is not None:
                return self._deferred
            if st[0] == "tdnn":
                _, l, relu, bn = st
                if relu and l.activation not in (None, "linear"):
                    raise ValueError("cannot fuse a ReLU after a TDNN that already has an activation")
                can_pool = (self.fuse_stats and not pooled and nxt is not None and nxt[0] == "stats" and
                            nxt[1].inputPeriod == 1 and l.units > 128 and l.padding == "SAME" and l.subsamplingFactor == 1)
                if not pooled:                           # (frame-level layers: behind the pooling everything is fp32 by design)
                    l.warn_fallback(gemm, relu)
            if gemm == L.GEMM_F16MX and st[0] == "tdnn" and not pooled and l.effective_gemm(gemm, relu) == gemm:
                # one half pass + two block-scaled residual passes (csrc/tdnn_mx.hip): activations travel as four chunk-major
                # planes (half value, e2m1 images of the residual and of the value, block scales) holding the ReLU outputs; a
                # layer's BatchNorm is folded into the weights of the next layer of the route
                if mxp is None:                          # first layer of the route: fp32 rows -> planes
                    B, T, D = x.shape
                    src = x if (x.dtype == torch.float32 and x.stride(2) == 1 and x.stride(0) == T * x.stride(1)) \
                        else x.to(torch.float32).contiguous()
                    mxp = Planes.buffers(self._ws.get, "mx_in", B, T, D, dev)
                    ops.mx_planes(src, D, lens, mxp)
                B, T, _ = mxp.shape
                fold, pending_bn = pending_bn, None
                plain = l.padding == "SAME" and l.subsamplingFactor == 1      # (VALID padding / subsampling: the 256-row kernel only)
                Tout = l.outputTimesteps(T)
                use_loader = plain and self._mx_use_loader(B, T)
                kern = "loader" if use_loader else "tile"
                wh, wq, bias = l.device_weights_mx(dev, fold=fold, kernel=kern)
                mxf = L.TDNN_MX_LOADER if use_loader else 0
                d = l.desc(gemm, torch.float32, torch.float32, act="relu" if relu else None, flags=mxf)      # (the MX entry points read no dtype field)
                nch_in = ops.round_up(l.inputDim, 32) // 32
                mx_flat = (self.mx_flat_rows and plain and not use_loader and B * T > 0 and B <= 4095 and B * T * nch_in * 64 < 2 ** 32
                           and B * T * (ops.round_up(l.units, 32) // 32) < 2 ** 31
                           and -(-(B * T) // 256) * 200 <= B * (-(-T // 256)) * 197)          # at least 1.5 % fewer tiles even if no frame was dropped
                if mx_flat:
                    row_starts = flat_rows_for(row_starts, B, T)
                if can_pool and mx_flat and self.flat_pooling:      # ... on flat row tiles (partial sums per run of an utterance's rows: flat_pooling)
                    sp = nxt[1]
                    od = 2 * l.units if sp.includeStd else l.units
                    slots = ops.flat_stats_slots(T) if self.deterministic else 0
                    sums = self._ws.get("sums", (B, max(slots, 1), 2, l.units), torch.float64, dev, padded=False)
                    sbuf = self._ws.get("pooled", (B, ops.round_up(od, 32)), torch.float32, dev)
                    scale, shift = bn.affine_device(dev) if bn is not None else (None, None)
                    d.flags = L.TDNN_DET_STATS if slots else 0
                    ops.tdnn_mx_flat_stats(mxp, row_starts, d, wh, wq, bias, scale, shift, sums, zero=not slots)
                    if slots:
                        ops.stats_finalize_flat(sums, row_starts, T, l.units, sp.includeStd, sp.epsilon, sbuf, slots)
                    else:
                        ops.stats_finalize(sums, lens, T, l.units, sp.includeStd, sp.epsilon, sbuf)
                    if si + 2 == tail_at:            # (the fused tail reads finished pooled rows: its own finalize knows per-utterance slots only)
                        self._deferred = DeferredTail(steps[tail_at][1], B, l.units, sp.includeStd, sp.epsilon, pooled=sbuf, lens=lens, T=T)
                    x = sbuf[:, :od].unsqueeze(0)
                    lens, pooled, skip, mxp = None, True, True, None
                    continue
                if can_pool:                             # ... -> reducing StatsPooling inside the epilogue (BatchNorm applied there)
                    sp = nxt[1]
                    od = 2 * l.units if sp.includeStd else l.units
                    slots = ops.stats_slots(T, mx_flags=mxf) if self.deterministic else 0
                    srows = ops.mx_slot_rows(mxf)
                    sums = self._ws.get("sums", (B, max(slots, 1), 2, l.units), torch.float64, dev, padded=False)
                    sbuf = self._ws.get("pooled", (B, ops.round_up(od, 32)), torch.float32, dev)
                    scale, shift = bn.affine_device(dev) if bn is not None else (None, None)
                    d.flags = mxf | (L.TDNN_DET_STATS if slots else 0)
                    ops.tdnn_mx_stats(mxp, lens, d, wh, wq, bias, scale, shift, sums, zero=not slots)
                    if si + 2 == tail_at:            # the caller's fused tail finalizes the sums itself
                        self._deferred = DeferredTail(steps[tail_at][1], B, l.units, sp.includeStd, sp.epsilon, sums=sums, slots=slots,
                                                      lens=lens, T=T, slot_rows=srows)
                    else:
                        ops.stats_finalize(sums, lens, T, l.units, sp.includeStd, sp.epsilon, sbuf, slots=slots, slot_rows=srows)
                    x = sbuf[:, :od].unsqueeze(0)
                    lens, pooled, skip, mxp = None, True, True, None
                    continue
                nl = nxt[1] if nxt is not None and nxt[0] == "tdnn" else None
                in_route = nl is not None and nl.effective_gemm(gemm, nxt[2]) == gemm and nl.inputDim == l.units
                defer_bn = bn is not None and in_route
                scale, shift = (None, None) if (bn is None or defer_bn) else bn.affine_device(dev)
                if in_route:
                    out = Planes.buffers(self._ws.get, out_role + "mx", B, Tout, l.units, dev)
                    if mx_flat:
                        ops.tdnn_mx_flat(mxp, row_starts, d, wh, wq, bias, scale, shift, out)
                    else:
                        ops.tdnn_mx(mxp, lens, d, wh, wq, bias, scale, shift, out)
                    mxp = out
                    x = out.xh                            # (shape carrier only)
                else:
                    ldy = ops.round_up(l.units, 32)
                    ybuf = self._ws.get(out_role, (B, Tout, ldy), torch.float32, dev, padded=ldy != l.units)
                    ops.tdnn_mx(mxp, lens, d, wh, wq, bias, scale, shift, ybuf)
                    mxp = None
                    x = ybuf[:, :, : l.units]
                if lens is not None and not plain:
                    lens, row_starts = ops.tdnn_out_lens(lens, d, torch.empty_like(lens)), None
                if defer_bn:
                    pending_bn = bn
                continue
            if mxp is not None:
                raise RuntimeError("internal: MX planes reached a layer that cannot read them")
            if pending_bn is not None:
                raise RuntimeError("internal: a deferred BatchNorm reached a layer that cannot fold it")
            if use_planes and st[0] == "tdnn" and not pooled and l.units > 128 and l.effective_gemm(gemm, relu) == gemm:
                if planes is None:                                   # first wide layer: split its fp32 input once
                    B, T, D = x.shape
                    planes = self._ws.get("split_in", (2, B, T, ops.round_up(D, 32)), torch.bfloat16, dev)
                    src = x if (x.dtype == torch.float32 and x.stride(2) == 1 and x.stride(0) == T * x.stride(1)) else x.to(torch.float32).contiguous()
                    ops.split_bf16(src, D, planes, lens)
                B, T = planes.shape[1], planes.shape[2]
                if can_pool:
                    flat = self.flat_pooling and self._flat_tiles(l, B, T, planes.shape[3])
                    if flat:
                        row_starts = flat_rows_for(row_starts, B, T)
                    x = self._pooled_by_gemm(l, relu, bn, nxt, planes, lens, gemm, True, dev, T,
                                             defer_to=steps[tail_at][1] if si + 2 == tail_at else None, row_starts=row_starts if flat else None)
                    lens, pooled, skip, planes = None, True, True, None
                    continue
                kint = bool(l.kernelWidth > 1)
                kflag = (L.TDNN_K_INTERLEAVED if kint else 0) | L.TDNN_W_TILED
                w, w_lo, bias = l.device_weights(dev, gemm, k_interleaved=kint, w_tiled=True)
                scale, shift = bn.affine_device(dev) if bn is not None else (None, None)
                Tout = l.outputTimesteps(T)
                ldy = ops.round_up(l.units, 32)
                out_lens = None
                if lens is not None and (l.padding == "VALID" or l.subsamplingFactor != 1):
                    out_lens = torch.empty_like(lens)
                keep = (nxt is not None and nxt[0] == "tdnn" and nxt[1].units > 128 and         # the consumer reads planes too
                        nxt[1].effective_gemm(gemm, nxt[2]) == gemm)
                # utterances that fill their 256-row tiles badly (a 1.5 s window: 148 rows): M-tiles over the batch's valid rows laid
                # end to end (ktf_tdnn_split_flat), same bits
                flat = self._flat_tiles(l, B, T, planes.shape[3])
                if flat:
                    row_starts = flat_rows_for(row_starts, B, T)
                split = (lambda d_, y_, ylo_: ops.tdnn_split_flat(planes, row_starts, d_, w, w_lo, bias, scale, shift, y_, ylo_)) if flat else \
                        (lambda d_, y_, ylo_: ops.tdnn_split(planes, lens, d_, w, w_lo, bias, scale, shift, y_, ylo_, out_lens))
                if keep:
                    ybuf = self._ws.get(out_role, (2, B, Tout, ldy), torch.bfloat16, dev, padded=ldy != l.units)
                    split(l.desc(gemm, torch.bfloat16, torch.bfloat16, act="relu" if relu else None, flags=kflag), ybuf[0], ybuf[1])
                    planes = ybuf
                    x = ybuf[0][:, :, : l.units]                     # shape carrier only (the values live in `planes`)
                else:
                    ybuf = self._ws.get(out_role, (B, Tout, ldy), torch.float32, dev, padded=ldy != l.units)
                    split(l.desc(gemm, torch.bfloat16, torch.float32, act="relu" if relu else None, flags=kflag), ybuf, None)
                    planes = None
                    x = ybuf[:, :, : l.units]
                if out_lens is not None:
                    lens, row_starts = out_lens, None
                continue
            if planes is not None:
                raise RuntimeError("internal: split planes reached a layer that cannot read them")
            if st[0] == "tdnn":
                if (can_pool and gemm in (L.GEMM_BF16, L.GEMM_BF16X3) and l.effective_gemm(gemm, relu) == gemm):
                    xdt = L.act_torch_dtype(gemm)
                    if x.dtype != xdt or x.stride(2) != 1 or x.stride(1) % 8 != 0 or x.stride(1) < ops.round_up(x.shape[-1], 32):
                        x = _padded_copy(x, xdt)
                    x = self._pooled_by_gemm(l, relu, bn, nxt, x, lens, gemm, False, dev, x.shape[1],
                                             defer_to=steps[tail_at][1] if si + 2 == tail_at else None)
                    lens, pooled, skip = None, True, True
                    continue
                g = L.GEMM_F32 if pooled else l.effective_gemm(gemm, relu)
                ydt = torch.float32 if (pooled or g != gemm) else act_dtype
                # the pair route (small batches of a reduced-precision model): this layer reads pairs if its producer wrote them,
                # and writes them if its consumer is a frame-level layer that can read them
                pair_in, x_pair = x_pair, False
                pair_ok = lambda t, r: t.activation in (None, "linear") or (t.activation == "relu" and not r)  # noqa: E731
                pair_out = (pairs and not pooled and g == L.GEMM_F32 and pair_ok(l, relu) and nxt is not None and nxt[0] == "tdnn"
                            and pair_ok(nxt[1], nxt[2]) and nxt[1].inputDim == l.units)
                if pair_in:
                    g = L.GEMM_BF16X4
                    if can_pool:                         # [affine, relu, batchnorm] -> reducing StatsPooling inside the pair kernel's epilogue
                        sp = nxt[1]
                        B, T, _ = x.shape
                        od = 2 * l.units if sp.includeStd else l.units
                        slots = ops.tdnn_stats_slots(T, g) if self.deterministic else 0
                        srows = ops.tdnn_slot_rows(g)
                        sums = self._ws.get("sums", (B, max(slots, 1), 2, l.units), torch.float64, dev, padded=False)
                        sbuf = self._ws.get("pooled", (B, ops.round_up(od, 32)), torch.float32, dev)
                        if x.stride(2) != 1 or x.stride(1) % 8 != 0 or x.stride(1) < ops.round_up(x.shape[-1], 32):
                            x = _padded_copy(x, torch.float32)
                        w, _, bias = l.device_weights(dev, g)
                        scale, shift = bn.affine_device(dev) if bn is not None else (None, None)
                        d = l.desc(g, L.PAIR, torch.float32, act="relu" if relu else None, flags=L.TDNN_DET_STATS if slots else 0)
                        ops.tdnn_stats(x, lens, d, w, None, bias, scale, shift, sums, zero=not slots)
                        if si + 2 == tail_at:            # the caller's fused tail finalizes the sums itself
                            self._deferred = DeferredTail(steps[tail_at][1], B, l.units, sp.includeStd, sp.epsilon, sums=sums, slots=slots,
                                                          lens=lens, T=T, slot_rows=srows)
                        else:
                            ops.stats_finalize(sums, lens, T, l.units, sp.includeStd, sp.epsilon, sbuf, slots=slots, slot_rows=srows)
                        x = sbuf[:, :od].unsqueeze(0)
                        lens, pooled, skip = None, True, True
                        continue
                if x.dtype != L.act_torch_dtype(g) or x.stride(2) != 1 or \
                        x.stride(1) % 8 != 0 or x.stride(1) < ops.round_up(x.shape[-1], 32):
                    x = _padded_copy(x, L.act_torch_dtype(g))
                B, T, _ = x.shape
                Tout = l.outputTimesteps(T)
                ldy = ops.round_up(l.units, 32)
                ybuf = self._ws.get(out_role, (B, Tout, ldy), ydt, dev, padded=ldy != l.units)
                out_lens = None
                if lens is not None and (l.padding == "VALID" or l.subsamplingFactor != 1):
                    out_lens = torch.empty_like(lens)
                sc_sh = bn.affine_device(dev) if bn is not None else None
                l.forward(x, lens=lens, relu=relu, bn=sc_sh, gemm=g, out_dtype=ydt, ldy=ldy, out=ybuf, out_lens=out_lens,
                          pair_in=pair_in, pair_out=pair_out)
                x_pair = pair_out
                if out_lens is not None:
                    lens, row_starts = out_lens, None
                x = ybuf[:, :, : l.units]
            elif st[0] == "stats":
                l = st[1]
                B, T, D = x.shape
                od = 2 * D if l.includeStd else D
                sbuf = self._ws.get("pooled", (B, ops.round_up(od, 32)), torch.float32, dev)
                l.reduce_all(x, D, lens=lens, out=sbuf)
                if si + 1 == tail_at:
                    return DeferredTail(steps[tail_at][1], B, D, l.includeStd, l.epsilon, pooled=sbuf, lens=lens, T=T)      # (lens: KTF_TAIL_SKIP_EMPTY of the
                                                                                                                         # short-utterance pass reads them)
                x = sbuf[:, :od].unsqueeze(0)       # (1, B, od): the pooled vectors form ONE B-row matrix
                lens = None
                pooled = True
            else:
                x = st[1](x.to(torch.float32).contiguous())
        if pooled:
            return x.reshape(x.shape[1], 1, x.shape[2])
        return x

    def __call__(self, inputs, training=False):
        x = inputs
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            x = ops.to_device_f32(x)
        if training:
            raise NotImplementedError("inference only")
        if any(not l.built for l in self.layers):
            self.input = Input(shape=(None, x.shape[-1]))
            self._build()
        if self._plan() is not None and x.dim() == 3:
            y = self.run_ragged(_padded_copy(x, torch.float32), None)
            return y.contiguous().clone()
        for l in self.layers:
            x = l(x)
        return x

    call = __call__


def _padded_copy(x, dtype):
    """(B,T,D) tensor/view -> view of a fresh (B,T,round_up(D,32)) buffer of `dtype` with zeroed pad columns."""
    B, T, D = x.shape
    Dp = ops.round_up(D, 32)
    src = x.contiguous()
    if src.dtype not in (torch.float32, torch.bfloat16) or (src.dtype != torch.float32 and src.dtype != dtype):
        src = src.to(torch.float32)
    dst = torch.empty((B, T, Dp), dtype=dtype, device=x.device)
    ops.convert_pad(src, D, dst)
    return dst[:, :, :D]


def SequentialFromConfig(cfg, nnet3Path=None, name=None, gemm="f32"):
    """models/kaldi/sequential.py:86-143 — `cfg["layers"]`: an "input" entry with `shape` [batch, time, feat], then
    entries for `cfg2layers`. With `nnet3Path`, every layer takes the weights of the nnet3 components matching its name
    (a layer without a match keeps its initialisation and is reported, as in the reference)."""
    entries = cfg.get("layers") or []
    if not entries:
        raise ValueError("no layers defined in config")
    head, body = entries[0], entries[1:]
    if head.get("type", "") != "input":
        raise ValueError("first layer in sequential model needs to be of type 'input'")
    batch, time, feat = head["shape"]
    stack = [Input(shape=(time, feat), batch_size=batch)]
    for entry in body:
        stack += cfg2layers(entry)
    mdl = Sequential(stack, name=name, gemm=gemm)
    if nnet3Path is None:
        return mdl
    source = KaldiNnet3Reader(nnet3Path, True)
    for layer in mdl.layers:
        try:
            weights = source.getWeights(layer.name)
        except KeyError:
            print(f"component with name '{layer.name}' not found in nnet3 model, skipping initialization")
            continue
        layer.set_weights(weights)
    return mdl


def downloadModel(link, outPath, sha256=None):
    """models/kaldi/download.py:28-100 fetches the Kaldi tarball; there is no network on the target machines, so
    only the "already present -> nothing to do" half of that contract is kept."""
    raise FileNotFoundError(
        f"pretrained Kaldi model not found under '{outPath}' and cannot be downloaded here (wanted {link}, "
        f"sha256 {sha256}); place the extracted tarball there")


def _load_yaml(path):
    import yaml
    with open(path, "r") as f:
        return yaml.safe_load(f)


def XvectorExtractorFromConfig(cfgPath, name=None, gemm="f32"):
    """models/kaldi/xvector_extractor.py:25-71 — `cfgPath`: YAML with an `extractor` section (framing / mfcc / vad / cmvn
    kwargs + `xvec` paths). The nnet3 model must already be on disk (see downloadModel)."""
    ext = _load_yaml(cfgPath)["extractor"]
    paths = ext["xvec"]
    if not os.path.exists(paths["model_path"]):
        kaldi = _load_yaml(paths["model_config_path"])
        target = os.path.join(os.path.dirname(paths["model_config_path"]), kaldi["name"])
        downloadModel(kaldi["download"]["link"], target, kaldi["download"]["hash"])
    return XvectorExtractor(ext, name=name, gemm=gemm)


class XvectorExtractor:
    """models/kaldi/xvector_extractor.py:74 — wav (batch, samples) in int16 scale -> length-normalised x-vector(s).

    The reference flattens the voiced frames of the whole batch into one sequence (:164-165) and is therefore only
    defined for batch = 1; here every batch row is an independent utterance and the result is (B, lda_dim)
    (squeezed to (lda_dim,) for B = 1 exactly like the reference's tf.squeeze).

    Scratch lives in per-stream workspaces (`_Workspace`); results handed to the caller are owned tensors."""

    def __init__(self, cfg, name=None, chunk_size=300, gemm="f32", **kwargs):
        nnet3 = _load_yaml(cfg["xvec"]["model_config_path"])
        seq = SequentialFromConfig(nnet3["model_config"], cfg["xvec"]["model_path"], "cmvn2xvec", gemm=gemm)
        self._setup(cfg, seq, ReadKaldiArray(cfg["xvec"]["global_mean_path"], binary=False),
                    ReadKaldiArray(cfg["xvec"]["lda_matrix_path"], binary=True), name)

    @classmethod
    def from_parts(cls, cfg, sequential, global_mean, lda_mat, name=None):
        """Build from an already-constructed Sequential and in-memory LDA parameters (used with synthetic weights)."""
        self = cls.__new__(cls)
        self._setup(cfg, sequential, global_mean, lda_mat, name)
        return self

    def _setup(self, cfg, sequential, global_mean, lda_mat, name):
        self.name = name if name is not None else "xvector_extractor"
        self.framing = Framing(**dict(cfg["framing"]))
        self.mfcc = MFCC(**cfg["mfcc"])
        self.vad = VAD(**cfg["vad"])
        self.cmvn = CMVN(**cfg["cmvn"])
        self.xvec = sequential
        self.gemm = sequential.gemm
        lda = np.asarray(lda_mat, np.float32)                          # transform.mat: (out, in + 1), last column = offset
        self.xvecGlobalMean = np.asarray(global_mean, np.float32)
        self.ldaOffset = np.ascontiguousarray(lda[..., -1:].T)          # (1, out)
        self.ldaMat = np.ascontiguousarray(lda[..., :-1].T)             # (in, out)
        self._post_dev = {}
        self._ws_own = _Workspace()
        self._tl = threading.local()      # per host thread: workspace override (compile), pinned short-utterance flag, last_lens / last_short_count
        self._graphs = {}
        self.fuse_tail = True        # pooling finalize + tdnn6 + mean-sub + LDA + length-norm as ONE launch (False: three, for A/B)
        self.fuse_tail_below = 512   # ... for batches below this size: from there on tdnn6 is a 256-workgroup fp32 MFMA GEMM over the
                                     # batch (0.10 ms at 1024 utterances, against 0.28 ms for the fused launch's vector arithmetic).
                                     # The exact fp32 mode keeps the one route at every size: its x-vectors do not depend on the
                                     # batch an utterance arrives in, bit for bit.
        self.route_short_utterances = True    # utterances with fewer voiced frames than the mode's Sequential.MIN_FRAMES go through the
                                              # tighter SHORT_MODE kernels, decided per utterance from the device's frame counts (_extract)
        self.verify_fraction = 0.0            # run-time guard of a reduced-precision mode on weights it was never tested on: this fraction of every
                                              # batch (at least one utterance) is extracted once more on the tighter SHORT_MODE kernels
                                              # (split-bf16: ~1e-5 from fp64) from the same features, and the largest difference is kept in
                                              # `last_verify` -- what the mode costs on THIS model and THIS audio (tolerance 1e-4;
                                              # tests/test_gpu_margin.py is where the shipped margin comes from). Eager calls only.
        self.verify_seed = 0x5EED
        self._warming_for_capture = False     # compile(): the warm-up calls run the short-utterance pass unconditionally (set and cleared by the
                                              # compiling thread; compile() is configuration, not a concurrent call: INTEGRATION.md)

    # ---- per-thread call state (see Sequential): `last_lens` / `last_short_count` are those of the CALLING thread's last call
    @property
    def _ws(self):
        return getattr(self._tl, "ws", None) or self._ws_own

    @_ws.setter
    def _ws(self, ws):
        self._tl.ws = None if ws is self._ws_own else ws

    @property
    def last_lens(self):
        """voiced-frame counts (B,) int32 of this thread's last call: a workspace view, valid until its next call"""
        return getattr(self._tl, "last_lens", None)

    @last_lens.setter
    def last_lens(self, v):
        self._tl.last_lens = v

    @property
    def last_short_count(self):
        """short utterances (below MIN_FRAMES voiced frames) of this thread's last eager call"""
        return getattr(self._tl, "last_short_count", 0)

    @last_short_count.setter
    def last_short_count(self, v):
        self._tl.last_short_count = v

    @property
    def last_verify(self):
        """{"n": utterances re-extracted, "max_abs_dev": largest |x-vector difference| against the tighter mode, "rows": their batch
        indices, "running_max": the largest seen by this thread so far} of this thread's last call with verify_fraction > 0 (else None)"""
        return getattr(self._tl, "last_verify", None)

    def _verify(self, feats, lens, y):
        """verify_fraction: a random subset of the batch through the SHORT_MODE kernels, compared with the rows of `y`."""
        seq = self.xvec
        B = feats.shape[0]
        mode = seq.SHORT_MODE.get(seq.gemm)
        if mode is None or B == 0:
            return
        tl = self._tl
        if getattr(tl, "verify_rng", None) is None:
            tl.verify_rng = np.random.default_rng(self.verify_seed + (threading.get_ident() & 0xFFFF))
        n = max(1, min(B, int(math.ceil(self.verify_fraction * B))))
        rows = np.sort(tl.verify_rng.choice(B, size=n, replace=False))
        idx = torch.as_tensor(rows, device=feats.device)
        sub_f = feats.index_select(0, idx).contiguous()
        sub_l = lens.index_select(0, idx).contiguous()
        y2 = self._xvectors(sub_f, sub_l, None, mode=mode)
        d = (y.index_select(0, idx) - y2).abs()
        d = torch.where(torch.isfinite(d), d, torch.zeros_like(d))           # (utterances without a voiced frame are NaN in both)
        dev = float(d.max().item())
        run = max(dev, (tl.last_verify or {}).get("running_max", 0.0)) if getattr(tl, "last_verify", None) else dev
        tl.last_verify = {"n": int(n), "max_abs_dev": dev, "rows": rows.tolist(), "mode": mode, "running_max": run}

    @property
    def layers(self):
        return [self.framing, self.mfcc, self.vad, self.cmvn, self.xvec]

    def _features(self, inputs):
        """wav -> (mfcc (B,T,C), CMVN'd voiced features (B,T,C) view, lens (B,)): workspace views."""
        fr, mf = self.framing, self.mfcc
        x, kind = fr.device_samples(inputs)          # fp32, or int16 PCM as it is
        if x.dim() == 1:
            x = x.unsqueeze(0)
        B, N = x.shape
        if N < fr.minSamples():
            raise ValueError(f"input sample size (axis=-1) must be >= frame size ({fr.frameSize})")
        if not mf.built or mf._M != fr.frameWidth:
            mf.build((None, None, fr.frameWidth))
        T = fr.numFrames(N)
        D = mf.numMfccs
        dev = x.device
        feat_dtype = L.act_torch_dtype(self.xvec.batch_gemm(B, T))
        ws = self._ws
        ws.enter(dev)
        mfcc = ws.get("mfcc", (B, T, D), torch.float32, dev, padded=False)
        feats = ws.get("feats", (B, T, ops.round_up(D, 32)), feat_dtype, dev, padded=D if D % 32 else False)
        lens = ws.get("lens", (B,), torch.int32, dev, padded=False)
        idx = ws.get("idx", (B, T), torch.int32, dev, padded=False)
        work = ws.get("cmvn_work", (B * T * 2 * D + 2 * D,), torch.float32, dev, padded=False)
        cfg = L.FrontendCfg.from_buffer_copy(mf._cfg)
        cfg.frame_size, cfg.frame_shift = fr.frameWidth, fr.frameShift
        cfg.pad_mode = 0 if fr.snipEdges else 1
        cfg.row_stride = 0 if x.is_contiguous() else x.stride(0)
        ops.frontend(x, kind, cfg, mf.tables(dev), L.OUT_MFCC, N, B, T, seed=mf.next_seed(), out=mfcc)
        ops.vad_cmvn(mfcc, self.vad.cfg(), self.cmvn.cfg(), feats, lens, idx, work)
        return mfcc, feats[:, :, :D], lens

    def features(self, inputs):
        """wav -> (mfcc (B,T,C), CMVN'd voiced features (B,T,C) with rows >= lens[b] unspecified, lens (B,) int32) — the
        front half of call(), as OWNED tensors."""
        mfcc, feats, lens = self._features(inputs)
        return mfcc.clone(), feats.clone(), lens.clone()

    def extract_stream(self, host_batches, depth=3):
        """Extension: x-vectors of a sequence of HOST batches (pinned (B,N) int16 / fp32 tensors) with the upload of batch
        i+1 on a separate HIP stream under the compute of batch i (`depth` device input buffers). Yields one (B, dim)
        device tensor per batch, in order. The compute path is the same as __call__; only the copies overlap."""
        L.require_gpu()
        dev = ops.default_device()
        compute = torch.cuda.current_stream(dev)
        copy = torch.cuda.Stream(device=dev)
        bufs, free = [None] * depth, [None] * depth
        for i, hb in enumerate(host_batches):
            hb = hb if isinstance(hb, torch.Tensor) else torch.as_tensor(hb)
            k = i % depth
            with torch.cuda.stream(copy):
                if free[k] is not None:
                    copy.wait_event(free[k])                      # the compute that read this buffer has finished
                if bufs[k] is None or bufs[k].shape != hb.shape or bufs[k].dtype != hb.dtype:
                    bufs[k] = torch.empty(hb.shape, dtype=hb.dtype, device=dev)
                bufs[k].copy_(hb, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(copy)
            compute.wait_event(ready)
            y = self(bufs[k])
            free[k] = torch.cuda.Event()
            free[k].record(compute)
            yield y

    def _extract(self, inputs, out=None):
        y, feats, lens = self._extract_routed(inputs, out)
        if (self.verify_fraction > 0 and self.gemm in self.xvec.SHORT_MODE and not torch.cuda.is_current_stream_capturing()
                and not self._warming_for_capture):
            self._verify(feats, lens, y)
        return y

    def _extract_routed(self, inputs, out=None):
        _, feats, lens = self._features(inputs)
        self.last_lens = lens                                          # voiced-frame counts of the last call (workspace view)
        B, T = feats.shape[0], feats.shape[1]
        seq = self.xvec
        nshort = seq.frames_floor(seq.gemm)
        if not (self.route_short_utterances and nshort > 0 and T >= nshort and seq.gemm in seq.SHORT_MODE
                and seq.batch_gemm(B, T) == _GEMM[seq.gemm]):
            return self._xvectors(feats, lens, out), feats, lens
        # Per-utterance routing: the batch runs in the model's mode with the utterances of fewer than `nshort` voiced frames masked out
        # (length 0: their tiles leave at once), then once more in the tighter mode with only those utterances live; the second tail
        # writes just their rows. The masks are made on the device (ktf_route_short, one small launch behind VAD / CMVN). Whether the
        # second pass is enqueued at all is decided on the host from ONE int32, the number of short utterances, which that kernel
        # writes straight into pinned host memory: the host reads it AFTER it has enqueued the whole first pass, i.e. while the GPU has
        # milliseconds of GEMMs queued -- no copy, no event, no bubble on the device, and a batch without short utterances (the usual
        # case) costs one 5 us launch. Under graph capture there is no host to ask: both passes are captured, and the second one's
        # workgroups leave at their first instruction when nothing is short.
        ws = self._ws
        lens_main = ws.get("lens_main", (B,), torch.int32, feats.device, padded=False)
        lens_short = ws.get("lens_short", (B,), torch.int32, feats.device, padded=False)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            ops.route_short(lens, nshort, lens_main, lens_short)
        else:
            tl = self._tl                                          # (a pinned flag and a sequence counter per calling thread)
            if getattr(tl, "short_flag", None) is None:
                tl.short_flag = torch.zeros(2, dtype=torch.int32).pin_memory()         # [count, sequence number]: written by the kernel
                tl.short_seq = 0
            tl.short_seq = (tl.short_seq % 0x3FFFFFFF) + 1
            ops.route_short(lens, nshort, lens_main, lens_short, tl.short_flag, tl.short_seq)
        y = self._xvectors(feats, lens_main, out)
        if not capturing:
            self.last_short_count = self._await_short_count(feats.device)
            if self.last_short_count == 0 and not self._warming_for_capture:
                return y, feats, lens
        short_mode = seq.SHORT_MODE[seq.gemm]
        if self._tail_fusable() and self.fuse_tail:
            self._xvectors(feats, lens_short, y, mode=short_mode, skip_empty=True)
        else:                                                          # (tails that write every row: select afterwards)
            y2 = self._xvectors(feats, lens_short, None, mode=short_mode)
            y.copy_(torch.where((lens_short > 0)[:, None], y2, y))
        return y, feats, lens

    def _await_short_count(self, dev):
        """The number of short utterances ktf_route_short wrote to pinned memory for this call: the host polls the sequence number
        (the GPU is busy with the first pass's launches meanwhile); after 2 s without it, one stream synchronisation."""
        import time
        flag, want = self._tl.short_flag, self._tl.short_seq
        t0 = time.perf_counter()
        spins = 0
        while int(flag[1]) != want:
            # The kernel sits behind everything this stream was given before it, so the wait can be as long as that work. The first
            # polls are back to back (a call on an idle stream has its answer within ~10 us); from then on the thread yields its
            # time slice and the interpreter lock between polls, then sleeps 50 us at a time: other host threads (extract_stream's
            # feeder, other ranks' Python on a shared core, concurrent extractors) run meanwhile instead of watching this one spin.
            spins += 1
            if spins > 200:
                time.sleep(0 if spins < 2000 else 50e-6)
            if time.perf_counter() - t0 > 2.0:
                torch.cuda.current_stream(dev).synchronize()
                if int(flag[1]) != want:
                    raise RuntimeError("ktf_route_short: the short-utterance count did not reach the host")
                break
        return int(flag[0])

    def _xvectors(self, feats, lens, out=None, mode=None, skip_empty=False):
        """CMVN'd features + voiced-frame counts -> x-vectors (B, lda_dim). `mode`: the TDNN arithmetic of this pass (default: the
        model's); `skip_empty`: utterances with lens == 0 are not computed and their rows of `out` stay (fused tail only)."""
        one_launch = self.fuse_tail and self._tail_fusable() and (skip_empty or feats.shape[0] < self.fuse_tail_below or
                                                                  self.xvec.batch_gemm(feats.shape[0], feats.shape[1], mode=mode) == L.GEMM_F32)
        if skip_empty and not one_launch:
            raise RuntimeError("internal: skip_empty needs the fused tail")
        h = self.xvec.run_ragged(feats, lens, defer_tail=one_launch, mode=mode)          # (B, 1, 512), or the deferred tail
        dev = feats.device
        key = str(dev)
        if key not in self._post_dev:
            self._post_dev[key] = (ops.to_device_f32(self.xvecGlobalMean, dev), ops.to_device_f32(self.ldaMat, dev),
                                   ops.to_device_f32(self.ldaOffset.reshape(-1), dev))
        mean, A, off = self._post_dev[key]
        if isinstance(h, DeferredTail):
            # pooling finalize + tdnn6 + mean subtraction + LDA + length normalisation in one launch (ktf_xvec_tail_f32)
            t = h
            if t.layer.units == A.shape[0]:
                w6, _, b6 = t.layer.device_weights(dev, L.GEMM_F32)
                B = t.B
                # utterances per workgroup: one while 64 workgroups per utterance still fill the chip, then groups (W is read once
                # per group); a large batch, or one with many slots per utterance (each of the 64 unit slices would add them up again:
                # 1.8 us per slot at B = 1), finalizes the pooled sums once in a launch of its own: same arithmetic either way
                group = max(1, min(32, (B * 64) // 2048))
                if t.sums is not None and (B > 8 or t.slots > 4):
                    sbuf = self.xvec._ws.get("pooled", (B, ops.round_up((2 if t.include_std else 1) * t.D, 32)), torch.float32, dev)
                    ops.stats_finalize(t.sums, t.lens, t.T, t.D, t.include_std, t.eps, sbuf, slots=t.slots, slot_rows=t.slot_rows)
                    t.pooled, t.sums = sbuf, None
                odim = A.shape[1]
                ws = self._ws
                partial = ws.get("tail_partial", (B, 64, odim), torch.float32, dev, padded=False)
                counters = ws.get("tail_cnt", (B,), torch.int32, dev, padded=True)      # zero when (re)allocated; the kernel leaves it zero
                if out is None:
                    out = torch.empty((B, odim), dtype=torch.float32, device=dev)
                return ops.xvec_tail(t.pooled, t.sums, t.slots, t.lens, t.T, t.D, t.include_std, t.eps, w6, b6, t.layer.units, mean, A, off,
                                     partial, counters, out, group=group, slot_rows=t.slot_rows, skip_empty=skip_empty)
            raise ValueError(f"LDA input dim {A.shape[0]} != embedding dim {t.layer.units}")
        B = h.shape[0]
        h2 = h.reshape(B, h.shape[-1])
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        return ops.xvec_post(h2, mean, A, off, out=out)

    def _tail_fusable(self):
        """ktf_xvec_tail_f32 serves LDA outputs up to 256 wide (one thread per output), an embedding layer whose width is the LDA's
        input, and default kernel flags; anything else keeps the three-launch tail (finalize, GEMM, ktf_xvec_post_f32)."""
        steps = self.xvec._plan() or []
        at = self.xvec._tail_step(steps)
        if at < 0:
            return False
        layer = steps[at][1]
        return self.ldaMat.shape[1] <= 256 and layer.units == self.ldaMat.shape[0] and layer.kernelFlags == 0

    def __call__(self, inputs, training=False):
        if hasattr(inputs, "shape") and len(inputs.shape) == 2 and inputs.shape[0] == 0:
            L.require_gpu()
            return torch.empty((0, self.ldaMat.shape[1]), dtype=torch.float32, device=ops.default_device())
        L.require_gpu()
        dev = inputs.device if (isinstance(inputs, torch.Tensor) and inputs.is_cuda) else ops.default_device()
        with L.launch_scope(dev):                        # device + stream looked up once for the ~10 launches of the call
            y = self._extract(inputs)
        return y.squeeze(0) if y.shape[0] == 1 else y

    call = __call__

    def compile(self, example):
        """Extension: capture the whole wav -> x-vector step for inputs of `example`'s shape / dtype into ONE HIP graph and
        return `run(wav) -> x-vector(s)`. A replay issues the same kernels on the same buffers as `__call__` (bit-identical
        results) with a single host call, which is what a latency-bound caller (batch 1: ~10 launches of 5-30 us each)
        wants. `run` copies `wav` into the captured input buffer, replays, and returns an owned tensor. Dither (a fresh
        seed per call) is frozen into the graph: compile a model whose MFCC has dither = 0."""
        L.require_gpu()
        ex, _ = self.framing.device_samples(example)
        if ex.dim() == 1:
            ex = ex.unsqueeze(0)
        static_in = ex.contiguous().clone()
        dev = static_in.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # The graph addresses scratch and weights by raw pointer. Scratch: PRIVATE workspaces for the capture (the model's own are
        # keyed by stream handle, and a later call on a recycled handle could grow -- i.e. free -- an arena the graph still
        # uses); they live as long as `run`. Weights: strong references to every device operand set the capture touched, and
        # the weights signature at capture time -- `run` refuses to replay after set_weights.
        own_ws, own_xws = self._ws, self.xvec._ws
        cap_ws, cap_xws = _Workspace(), _Workspace()
        self._ws, self.xvec._ws = cap_ws, cap_xws
        try:
            with torch.cuda.stream(side):                # warm-up on the capture stream: workspaces, tables, LDS opt-ins -- of BOTH
                self._warming_for_capture = True         # passes of a routed batch (the capture contains the short-utterance pass
                try:                                     # whether or not the example has a short utterance; its weights cannot be
                    for _ in range(2):                   # uploaded while the stream is capturing)
                        self._extract(static_in)
                finally:
                    self._warming_for_capture = False
                static_out = torch.empty((static_in.shape[0], self.ldaMat.shape[1]), dtype=torch.float32, device=dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    self._extract(static_in, out=static_out)
        finally:
            self._ws, self.xvec._ws = own_ws, own_xws
        torch.cuda.current_stream(dev).wait_stream(side)
        sig = self.xvec.weights_signature()
        keep = [cap_ws, cap_xws, dict(self._post_dev)]
        for l in self.xvec.layers:
            if isinstance(l, TDNN):
                keep.append(dict(l._dev))
            elif isinstance(l, BatchNorm):
                keep.append(l._dev)

        epoch = [WEIGHTS_EPOCH[0]]

        def run(wav):
            if WEIGHTS_EPOCH[0] != epoch[0]:
                # some layer in the process changed: is it one of ours? (slow path, only after a set_weights somewhere)
                if self.xvec.weights_signature() != sig:
                    raise RuntimeError("the model's weights changed after compile(): capture again")
                epoch[0] = WEIGHTS_EPOCH[0]
            w, _ = self.framing.device_samples(wav)
            if w.dim() == 1:
                w = w.unsqueeze(0)
            if w.shape != static_in.shape or w.dtype != static_in.dtype:
                raise ValueError(f"compiled for {tuple(static_in.shape)} {static_in.dtype}, got {tuple(w.shape)} {w.dtype}")
            static_in.copy_(w)
            graph.replay()
            y = static_out.clone()
            return y.squeeze(0) if y.shape[0] == 1 else y

        run.graph = graph
        run.keepalive = keep
        return run
