"""
ktf.models — SequentialFromConfig / XvectorExtractor / XvectorExtractorFromConfig with the
reference's signatures (kaldi_tflite/lib/models/kaldi/{sequential,xvector_extractor}.py).

The builders parse the same two-level YAML as the reference. At call time the model does not
run layer by layer: the canonical chains are recognised and dispatched to fused kernels
(Framing+MFCC in one launch, VAD+compaction+CMVN in one launch, every [affine, relu, batchnorm]
triple as one MFMA GEMM with a fused epilogue), on ragged utterance-strided activations so a
batch of B utterances equals B independent batch-1 calls of the reference.
"""

import os

import numpy as np
import torch

from . import _lib as L
from . import ops
from .io import KaldiNnet3Reader, ReadKaldiArray
from .layers import TDNN, BatchNorm, CMVN, Framing, Layer, MFCC, ReLU, StatsPooling, VAD, _GEMM


class Input:
    """Placeholder for keras.layers.Input: carries the declared (batch, time, feat) shape."""

    def __init__(self, shape=None, batch_size=None, name="input"):
        self.shape = (batch_size,) + tuple(shape)
        self.name = name


def cfg2layers(layerCfg):
    """models/kaldi/sequential.py:29-83: one config entry -> list of layers."""
    layerTypes = layerCfg.get("type", [])
    if isinstance(layerTypes, str):
        layerTypes = [layerTypes]
    if len(layerTypes) == 0:
        raise KeyError("layer config does not define layer 'type'")
    name = layerCfg.get("name", None)
    layers = []
    for layerType in layerTypes:
        t = layerType.lower()
        cfg = layerCfg.get("cfg", {})
        if t in ["affine", "tdnn"]:
            cfg["name"] = f"{name}.affine"
            layer = TDNN(**cfg)
        elif t in ["relu"]:
            layer = ReLU(name=f"{name}.relu")
        elif t in ["batchnorm", "bn"]:
            layer = BatchNorm(name=f"{name}.batchnorm")
        elif t in ["stats", "stats_extraction", "stats_pooling"]:
            cfg["name"] = name
            layer = StatsPooling(**cfg)
        else:
            raise ValueError(f"unsupported layer type '{t}'")
        layers.append(layer)
    return layers


class Sequential:
    """Keras-Sequential stand-in: `.layers`, `.name`, `mdl(x, training=False)`, `get_layer`, `summary`.
    `gemm` selects the TDNN arithmetic of the fused runner: "f32" (exact fp32 MFMA, default = the reference's
    precision), "bf16" or "bf16x3"."""

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
        self.fuse_stats = True      # bf16 mode: pool inside the epilogue of the GEMM that feeds a reducing StatsPooling
        self.dtype = "float32"
        self._ws = {}
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
        while i < len(Ls):
            l = Ls[i]
            if isinstance(l, TDNN):
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
                i += 1
            elif isinstance(l, (ReLU, BatchNorm)):
                steps.append(("eltwise", l))
                i += 1
            else:
                return None
        return steps

    split_planes = os.environ.get("KTF_X3_SPLIT", "1") != "0"     # bf16x3: hi/lo activation planes between wide layers
    # batches with fewer 256-row tiles than this run on the exact fp32 kernels (crossover of the measured per-layer times)
    min_tiles = {"bf16": 6, "f16": 6, "bf16x3": 32}

    def batch_gemm(self, B, T):
        """GEMM arithmetic for a batch of B utterances of up to T frames: the model's mode, except that a handful of
        256-row tiles (single utterances) cannot fill the chip on the 256-wide ring kernels -- the exact fp32 kernels have
        small-tile forms and are faster there (one 10 s utterance: 0.19 ms against 0.20 bf16 / 0.39 split-bf16)."""
        gemm = _GEMM[self.gemm]
        if gemm != L.GEMM_F32 and B * ((T + 255) // 256) < self.min_tiles.get(self.gemm, 0):
            gemm = L.GEMM_F32
        return gemm

    def run_ragged(self, x, lens=None):
        """x: (B, T, D) view of an utterance-strided buffer whose row stride is a multiple of 8 and >= round_up(D, 32)
        (pad columns finite); lens: int32 (B,) valid rows per utterance or None. Returns (B, T', units) for frame-level
        outputs or (B, 1, units) after a reducing StatsPooling."""
        steps = self._plan()
        if steps is None:
            raise NotImplementedError("this layer stack is not supported by the fused ragged runner")
        gemm = self.batch_gemm(x.shape[0], x.shape[1])
        act_dtype = L.act_torch_dtype(gemm)
        pooled = False
        skip = False
        # split-bf16 mode: frame-level activations travel between the wide layers as hi/lo bf16 planes (2,B,T,ld) instead
        # of fp32, so the GEMM K-loop carries no conversion (ktf_tdnn_split); `planes` holds them while they exist
        use_planes = gemm == L.GEMM_BF16X3 and self.split_planes
        planes = None
        for si, st in enumerate(steps):
            if skip:
                skip = False
                continue
            nxt = steps[si + 1] if si + 1 < len(steps) else None
            if use_planes and st[0] == "tdnn" and not pooled and st[1].units > 128:
                _, l, relu, bn = st
                if relu and l.activation not in (None, "linear"):
                    raise ValueError("cannot fuse a ReLU after a TDNN that already has an activation")
                if planes is None:                                   # first wide layer: split its fp32 input once
                    B, T, D = x.shape
                    planes = self._buffer(("xp", id(l), B, T, str(x.device)), (2, B, T, ops.round_up(D, 32)), torch.bfloat16, x.device)
                    src = x if (x.dtype == torch.float32 and x.stride(2) == 1 and x.stride(0) == T * x.stride(1)) else x.to(torch.float32).contiguous()
                    ops.split_bf16(src, D, planes)
                B, T = planes.shape[1], planes.shape[2]
                w, w_lo, bias = l.device_weights(x.device, gemm)
                scale, shift = bn.affine_device(x.device) if bn is not None else (None, None)
                fuse = (self.fuse_stats and nxt is not None and nxt[0] == "stats" and nxt[1].inputPeriod == 1 and
                        l.padding == "SAME" and l.subsamplingFactor == 1)
                if fuse:
                    sp = nxt[1]
                    D = l.units
                    od = 2 * D if sp.includeStd else D
                    ld = ops.round_up(od, 32)
                    sums = self._buffer(("sum", id(l), B, D, str(x.device)), (B, 2, D), torch.float64, x.device)
                    sbuf = self._buffer(("s", id(sp), B, ld, str(x.device)), (B, ld), torch.float32, x.device)
                    d = l.desc(gemm, torch.bfloat16, torch.bfloat16, act="relu" if relu else None)
                    ops.tdnn_split_stats(planes, lens, d, w, w_lo, bias, scale, shift, sums)
                    ops.stats_finalize(sums, lens, T, D, sp.includeStd, sp.epsilon, sbuf)
                    x = sbuf[:, :od].unsqueeze(0)
                    lens, pooled, skip, planes = None, True, True, None
                    continue
                Tout = l.outputTimesteps(T)
                ldy = ops.round_up(l.units, 32)
                out_lens = None
                if lens is not None and (l.padding == "VALID" or l.subsamplingFactor != 1):
                    out_lens = torch.empty_like(lens)
                keep = nxt is not None and nxt[0] == "tdnn" and nxt[1].units > 128        # the consumer reads planes too
                if keep:
                    ybuf = self._buffer(("yp", id(l), B, Tout, ldy, str(x.device)), (2, B, Tout, ldy), torch.bfloat16, x.device)
                    d = l.desc(gemm, torch.bfloat16, torch.bfloat16, act="relu" if relu else None)
                    ops.tdnn_split(planes, lens, d, w, w_lo, bias, scale, shift, ybuf[0], ybuf[1], out_lens)
                    planes = ybuf
                    x = ybuf[0][:, :, : l.units]                     # shape carrier only (the values live in `planes`)
                else:
                    ybuf = self._buffer(("y", id(l), B, Tout, ldy, torch.float32, str(x.device)), (B, Tout, ldy), torch.float32, x.device)
                    d = l.desc(gemm, torch.bfloat16, torch.float32, act="relu" if relu else None)
                    ops.tdnn_split(planes, lens, d, w, w_lo, bias, scale, shift, ybuf, None, out_lens)
                    planes = None
                    x = ybuf[:, :, : l.units]
                if out_lens is not None:
                    lens = out_lens
                continue
            if planes is not None:
                raise RuntimeError("internal: split planes reached a layer that cannot read them")
            if (self.fuse_stats and st[0] == "tdnn" and not pooled and gemm in (L.GEMM_BF16, L.GEMM_BF16X3, L.GEMM_F16) and nxt is not None
                    and st[1].effective_gemm(gemm, st[2]) == gemm
                    and nxt[0] == "stats" and nxt[1].inputPeriod == 1 and st[1].units > 128 and st[1].padding == "SAME"
                    and st[1].subsamplingFactor == 1):
                # [affine, relu, batchnorm] -> reducing StatsPooling: pooled inside the GEMM epilogue, y is never written
                _, l, relu, bn = st
                sp = nxt[1]
                xdt = L.act_torch_dtype(gemm)
                if x.dtype != xdt or x.stride(2) != 1 or x.stride(1) % 8 != 0 or x.stride(1) < ops.round_up(x.shape[-1], 32):
                    x = _padded_copy(x, xdt)
                B, T, _ = x.shape
                D = l.units
                od = 2 * D if sp.includeStd else D
                ld = ops.round_up(od, 32)
                sums = self._buffer(("sum", id(l), B, D, str(x.device)), (B, 2, D), torch.float64, x.device)
                sbuf = self._buffer(("s", id(sp), B, ld, str(x.device)), (B, ld), torch.float32, x.device)
                w, w_lo, bias = l.device_weights(x.device, gemm)
                d = l.desc(gemm, x.dtype, xdt, act="relu" if relu else None)
                scale, shift = bn.affine_device(x.device) if bn is not None else (None, None)
                ops.tdnn_stats(x, lens, d, w, w_lo, bias, scale, shift, sums)
                ops.stats_finalize(sums, lens, T, D, sp.includeStd, sp.epsilon, sbuf)
                x = sbuf[:, :od].unsqueeze(0)
                lens = None
                pooled = True
                skip = True
                continue
            if st[0] == "tdnn":
                _, l, relu, bn = st
                g = L.GEMM_F32 if pooled else l.effective_gemm(gemm, relu)
                ydt = torch.float32 if (pooled or g != gemm) else act_dtype
                if x.dtype != L.act_torch_dtype(g) or x.stride(2) != 1 or \
                        x.stride(1) % 8 != 0 or x.stride(1) < ops.round_up(x.shape[-1], 32):
                    x = _padded_copy(x, L.act_torch_dtype(g))
                B, T, _ = x.shape
                Tout = l.outputTimesteps(T)
                ldy = ops.round_up(l.units, 32)
                ybuf = self._buffer(("y", id(l), B, Tout, ldy, ydt, str(x.device)), (B, Tout, ldy), ydt, x.device)
                out_lens = None
                if lens is not None and (l.padding == "VALID" or l.subsamplingFactor != 1):
                    out_lens = torch.empty_like(lens)
                sc_sh = bn.affine_device(x.device) if bn is not None else None
                l.forward(x, lens=lens, relu=relu, bn=sc_sh, gemm=g, out_dtype=ydt, ldy=ldy, out=ybuf, out_lens=out_lens)
                if out_lens is not None:
                    lens = out_lens
                x = ybuf[:, :, : l.units]
            elif st[0] == "stats":
                l = st[1]
                B, T, D = x.shape
                od = 2 * D if l.includeStd else D
                ld = ops.round_up(od, 32)
                sbuf = self._buffer(("s", id(l), B, ld, str(x.device)), (B, ld), torch.float32, x.device)
                l.reduce_all(x, D, lens=lens, out=sbuf)
                x = sbuf[:, :od].unsqueeze(0)       # (1, B, od): the pooled vectors form ONE B-row matrix
                lens = None
                pooled = True
            else:
                l = st[1]
                xc = x.to(torch.float32).contiguous()
                x = l(xc)
        if pooled:
            return x.reshape(x.shape[1], 1, x.shape[2])
        return x

    def _buffer(self, key, shape, dtype, device):
        b = self._ws.get(key)
        if b is None:
            b = torch.zeros(shape, dtype=dtype, device=device)   # zeros: pad columns must stay finite
            self._ws[key] = b
        return b

    def __call__(self, inputs, training=False):
        x = inputs
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            x = ops.to_device_f32(x)
        if training:
            raise NotImplementedError("inference only")
        for l in self.layers:
            if not l.built:
                self.input = Input(shape=(None, x.shape[-1]))
                self._build()
                break
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
    if src.dtype not in (torch.float32, torch.bfloat16, torch.float16) or (src.dtype != torch.float32 and src.dtype != dtype):
        src = src.to(torch.float32)
    dst = torch.empty((B, T, Dp), dtype=dtype, device=x.device)
    ops.convert_pad(src, D, dst)
    return dst[:, :, :D]


def SequentialFromConfig(cfg, nnet3Path=None, name=None, gemm="f32"):
    """models/kaldi/sequential.py:86-143."""
    layersConfig = cfg.get("layers", [])
    if len(layersConfig) == 0:
        raise ValueError("no layers defined in config")
    inputCfg = layersConfig[0]
    if inputCfg.get("type", "") != "input":
        raise ValueError("first layer in sequential model needs to be of type 'input'")
    batchSize, timesteps, featDim = inputCfg["shape"]
    layers = [Input(shape=(timesteps, featDim), batch_size=batchSize)]
    for lCfg in cfg["layers"][1:]:
        layers.extend(cfg2layers(lCfg))
    mdl = Sequential(layers, name=name, gemm=gemm)
    if nnet3Path is not None:
        nnet3Mdl = KaldiNnet3Reader(nnet3Path, True)
        for layer in mdl.layers:
            try:
                layer.set_weights(nnet3Mdl.getWeights(layer.name))
            except KeyError:
                print(f"component with name '{layer.name}' not found in nnet3 model, skipping initialization")
    return mdl


def downloadModel(link, outPath, sha256=None):
    """models/kaldi/download.py:28-100 fetches the Kaldi tarball; there is no network on the target machines, so
    only the "already present -> nothing to do" half of that contract is kept."""
    raise FileNotFoundError(
        f"pretrained Kaldi model not found under '{outPath}' and cannot be downloaded here (wanted {link}, "
        f"sha256 {sha256}); place the extracted tarball there")


def XvectorExtractorFromConfig(cfgPath, name=None, gemm="f32"):
    """models/kaldi/xvector_extractor.py:25-71."""
    import yaml
    with open(cfgPath) as f:
        cfg = yaml.safe_load(f)
    with open(cfg["extractor"]["xvec"]["model_config_path"], "r") as f:
        kaldiCfg = yaml.safe_load(f)
    kaldiMdlPath = cfg["extractor"]["xvec"]["model_path"]
    if not os.path.exists(kaldiMdlPath):
        downloadDir = os.path.join(os.path.dirname(cfg["extractor"]["xvec"]["model_config_path"]), kaldiCfg["name"])
        downloadModel(kaldiCfg["download"]["link"], downloadDir, kaldiCfg["download"]["hash"])
    return XvectorExtractor(cfg["extractor"], name=name, gemm=gemm)


class XvectorExtractor:
    """models/kaldi/xvector_extractor.py:74 — wav (batch, samples) in int16 scale -> length-normalised x-vector(s).

    The reference flattens the voiced frames of the whole batch into one sequence (:164-165) and is therefore only
    defined for batch = 1; here every batch row is an independent utterance and the result is (B, lda_dim)
    (squeezed to (lda_dim,) for B = 1 exactly like the reference's tf.squeeze)."""

    def __init__(self, cfg, name=None, chunk_size=300, gemm="f32", **kwargs):
        import yaml
        self.name = name if name is not None else "xvector_extractor"
        fcfg = dict(cfg["framing"])
        self.framing = Framing(**fcfg)
        self.mfcc = MFCC(**cfg["mfcc"])
        self.vad = VAD(**cfg["vad"])
        self.cmvn = CMVN(**cfg["cmvn"])
        with open(cfg["xvec"]["model_config_path"], "r") as f:
            nnet3Cfg = yaml.safe_load(f)
        self.xvec = SequentialFromConfig(nnet3Cfg["model_config"], cfg["xvec"]["model_path"], "cmvn2xvec", gemm=gemm)
        globalMean = ReadKaldiArray(cfg["xvec"]["global_mean_path"], binary=False)
        ldaMat = ReadKaldiArray(cfg["xvec"]["lda_matrix_path"], binary=True)
        self._init_post(globalMean, ldaMat)
        self.gemm = gemm
        self._ws = {}

    @classmethod
    def from_parts(cls, cfg, sequential, global_mean, lda_mat, name=None):
        """Build from an already-constructed Sequential and in-memory LDA parameters (used with synthetic weights)."""
        self = cls.__new__(cls)
        self.name = name if name is not None else "xvector_extractor"
        self.framing = Framing(**cfg["framing"])
        self.mfcc = MFCC(**cfg["mfcc"])
        self.vad = VAD(**cfg["vad"])
        self.cmvn = CMVN(**cfg["cmvn"])
        self.xvec = sequential
        self._init_post(global_mean, lda_mat)
        self.gemm = sequential.gemm
        self._ws = {}
        return self

    def _init_post(self, globalMean, ldaMat):
        ldaMat = np.asarray(ldaMat, np.float32)
        self.xvecGlobalMean = np.asarray(globalMean, np.float32)
        self.ldaOffset = np.ascontiguousarray(ldaMat[..., -1:].T)      # (1, out)
        self.ldaMat = np.ascontiguousarray(ldaMat[..., :-1].T)         # (in, out)
        self._post_dev = None

    @property
    def layers(self):
        return [self.framing, self.mfcc, self.vad, self.cmvn, self.xvec]

    def _workspace(self, B, T, D, device, feat_dtype):
        key = (B, T, D, str(device), feat_dtype)
        ws = self._ws.get(key)
        if ws is None:
            ld = ops.round_up(D, 32)
            ws = {
                "mfcc": torch.empty((B, T, D), dtype=torch.float32, device=device),
                "feats": torch.zeros((B, T, ld), dtype=feat_dtype, device=device),
                "lens": torch.zeros((B,), dtype=torch.int32, device=device),
                "idx": torch.empty((B, T), dtype=torch.int32, device=device),
                "work": torch.empty((B * T * 2 * D + 2 * D,), dtype=torch.float32, device=device),
            }
            self._ws[key] = ws
        return ws

    def features(self, inputs):
        """wav -> (mfcc (B,T,C), cmvn'd voiced features view (B,T,C), lens (B,)) — the front half of call()."""
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
        gemm = self.xvec.batch_gemm(B, T)
        feat_dtype = L.act_torch_dtype(gemm)
        ws = self._workspace(B, T, D, x.device, feat_dtype)
        cfg = L.FrontendCfg.from_buffer_copy(mf._cfg)
        cfg.frame_size, cfg.frame_shift = fr.frameWidth, fr.frameShift
        cfg.pad_mode = 0 if fr.snipEdges else 1
        cfg.row_stride = 0 if x.is_contiguous() else x.stride(0)
        ops.frontend(x, kind, cfg, mf.tables(x.device), L.OUT_MFCC, N, B, T, seed=mf.next_seed(), out=ws["mfcc"])
        ops.vad_cmvn(ws["mfcc"], self.vad.cfg(), self.cmvn.cfg(), ws["feats"], ws["lens"], ws["idx"], ws["work"])
        return ws["mfcc"], ws["feats"][:, :, :D], ws["lens"]

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

    def __call__(self, inputs, training=False):
        if hasattr(inputs, "shape") and len(inputs.shape) == 2 and inputs.shape[0] == 0:
            L.require_gpu()
            return torch.empty((0, self.ldaMat.shape[1]), dtype=torch.float32, device=ops.default_device())
        _, feats, lens = self.features(inputs)
        h = self.xvec.run_ragged(feats, lens)                      # (B, 1, 512)
        B = h.shape[0]
        h2 = h.reshape(B, h.shape[-1])
        if self._post_dev is None or self._post_dev[0].device != h2.device:
            self._post_dev = (ops.to_device_f32(self.xvecGlobalMean, h2.device), ops.to_device_f32(self.ldaMat, h2.device),
                              ops.to_device_f32(self.ldaOffset.reshape(-1), h2.device))
        mean, A, off = self._post_dev
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        y = ops.xvec_post(h2, mean, A, off)
        return y.squeeze(0) if B == 1 else y

    call = __call__
