"""
ctypes binding of libktf_hip.so (the C-ABI declared in include/ktf_hip.h).

This is the only place the Python host side touches native code. There is NO CPU
fallback: if the shared library is missing or no MI355X is visible, every compute
entry point raises (`KtfBackendError`) instead of silently computing elsewhere.
"""

import ctypes as C
import threading
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree library. KTF_LIBRARY names another build of the same ABI (probe / timing-ablation builds of
# tools/) and is honoured ONLY together with KTF_ALLOW_LIBRARY_OVERRIDE=1, so that a variable left over in a shell cannot
# silently swap the library; bench.py records the path that was loaded.
_OVERRIDE = os.environ.get("KTF_LIBRARY") if os.environ.get("KTF_ALLOW_LIBRARY_OVERRIDE") == "1" else None
LIB_PATH = _OVERRIDE or os.path.join(_HERE, "libktf_hip.so")

KTF_F32, KTF_BF16, KTF_BF16P = 0, 1, 3
PAIR = "bf16p"                    # stands for KTF_BF16P where a torch dtype is expected: pairs live in float32 tensors
GEMM_F32, GEMM_BF16, GEMM_BF16X3, GEMM_F16MX, GEMM_BF16X4 = 0, 1, 2, 5, 6
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
(ACT_ELU, ACT_SELU, ACT_SOFTPLUS, ACT_SOFTSIGN, ACT_SWISH, ACT_GELU, ACT_EXPONENTIAL, ACT_HARD_SIGMOID,
 ACT_SOFTMAX) = range(4, 13)         # run as a pass of their own (ktf_activation_f32)
TDNN_REF_TILES, TDNN_DET_STATS, TDNN_K_INTERLEAVED, TDNN_W_TILED = 1, 2, 4, 8   # KtfTdnnDesc.flags
TAIL_SKIP_EMPTY = 1                 # ktf_xvec_tail_f32 flags
TDNN_MX_LOADER = 1 << 24          # ktf_tdnn_mx*: the loader-wave kernel (csrc/tdnn_mxl.hip) and its weight images

IN_WAV, IN_FRAMES, IN_WINDOWED, IN_WAV_I16 = 0, 1, 2, 3
OUT_FRAMES, OUT_WINDOWED, OUT_FBANK, OUT_MFCC = 0, 1, 2, 3


class KtfBackendError(RuntimeError):
    pass


def ktf_dtype(t):
    """torch dtype -> KTF_* element type of an activation / weight buffer."""
    import torch
    return {torch.float32: KTF_F32, torch.bfloat16: KTF_BF16, PAIR: KTF_BF16P}[t]


def act_torch_dtype(gemm):
    """Storage dtype of the frame-level activations for a GEMM mode."""
    import torch
    return {GEMM_BF16: torch.bfloat16}.get(gemm, torch.float32)


class FrontendCfg(C.Structure):
    _fields_ = [
        ("frame_size", C.c_int32), ("frame_shift", C.c_int32), ("nfft", C.c_int32), ("num_mels", C.c_int32),
        ("num_ceps", C.c_int32), ("remove_dc", C.c_int32), ("raw_energy", C.c_int32), ("use_energy", C.c_int32),
        ("use_power", C.c_int32), ("use_log", C.c_int32), ("use_lifter", C.c_int32), ("preemph", C.c_float),
        ("dither", C.c_float), ("energy_floor", C.c_float), ("eps", C.c_float), ("pad_mode", C.c_int32), ("row_stride", C.c_int32),
    ]


class FrontendTables(C.Structure):
    _fields_ = [
        ("window", C.c_void_p), ("twiddle", C.c_void_p), ("rtwiddle", C.c_void_p), ("mel_start", C.c_void_p),
        ("mel_len", C.c_void_p), ("mel_w", C.c_void_p), ("dct", C.c_void_p), ("lifter", C.c_void_p),
        ("fast_tw", C.c_void_p), ("fast_mel_meta", C.c_void_p), ("fast_mel_w", C.c_void_p),
        ("mel_stride", C.c_int32), ("reserved", C.c_int32),
    ]


class VadCfg(C.Structure):
    _fields_ = [("energy_threshold", C.c_float), ("energy_mean_scale", C.c_float), ("proportion_threshold", C.c_float),
                ("frames_context", C.c_int32), ("energy_coeff", C.c_int32)]


class CmvnCfg(C.Structure):
    _fields_ = [("window", C.c_int32), ("norm_vars", C.c_int32), ("valid", C.c_int32), ("reserved", C.c_int32)]


class TdnnDesc(C.Structure):
    _fields_ = [("units", C.c_int32), ("din", C.c_int32), ("din_pad", C.c_int32), ("nctx", C.c_int32),
                ("ctx", C.c_int32 * 16), ("subsampling", C.c_int32), ("valid", C.c_int32), ("act", C.c_int32),
                ("gemm", C.c_int32), ("x_dtype", C.c_int32), ("w_dtype", C.c_int32), ("y_dtype", C.c_int32),
                ("flags", C.c_int32)]


_P = C.c_void_p
_i64, _i32, _f32, _u64 = C.c_int64, C.c_int32, C.c_float, C.c_uint64

# name -> (restype, argtypes); mirrors include/ktf_hip.h one to one
PROTOTYPES = {
    "ktf_version": (_i32, []),
    "ktf_last_error": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "ktf_build_id": (C.c_char_p, []),
    "ktf_clock_probe": (C.c_int, [_P, _i64, _P]),
    "ktf_num_frames": (_i64, [_i64, _i32, _i32]),
    "ktf_num_frames_padded": (_i64, [_i64, _i32, _i32, _i32]),
    "ktf_frontend_f32": (C.c_int, [_P, _i64, _i64, _i32, C.POINTER(FrontendCfg), C.POINTER(FrontendTables), _i32, _P, _P, _u64, _P]),
    "ktf_dct_f32": (C.c_int, [_P, _i64, _i32, _i32, _P, _P, _P, _P]),
    "ktf_vad_mask_f32": (C.c_int, [_P, _i64, _i64, _i32, C.POINTER(VadCfg), _P, _P]),
    "ktf_vad_index": (C.c_int, [_P, _i64, _i64, _i32, C.POINTER(VadCfg), _P, _P, _P]),
    "ktf_cmvn_f32": (C.c_int, [_P, _i64, _i64, _i32, _i64, _P, C.POINTER(CmvnCfg), _P, _i64, _P, _P, _P]),
    "ktf_vad_cmvn": (C.c_int, [_P, _i64, _i64, _i32, C.POINTER(VadCfg), C.POINTER(CmvnCfg), _P, _i32, _i64, _P, _P, _P, _P]),
    "ktf_route_short": (C.c_int, [_P, _i64, _i32, _P, _P, _P, _i32, _P]),
    "ktf_tdnn_out_len": (_i64, [_i64, C.POINTER(TdnnDesc)]),
    "ktf_tdnn_out_lens": (C.c_int, [_P, _i64, C.POINTER(TdnnDesc), _P, _P]),
    "ktf_tdnn_last_kernel": (C.c_char_p, []),
    "ktf_tdnn": (C.c_int, [_P, _i64, _i64, _i64, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _i64, _P, _P]),
    "ktf_tdnn_stats": (C.c_int, [_P, _i64, _i64, _i64, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ktf_tdnn_split": (C.c_int, [_P, _P, _i64, _i64, _i64, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P, _i64, _P, _P]),
    "ktf_tdnn_split_flat": (C.c_int, [_P, _P, _i64, _i64, _i64, _P, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P, _i64, _P]),
    "ktf_tdnn_split_stats": (C.c_int, [_P, _P, _i64, _i64, _i64, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ktf_tdnn_split_flat_stats": (C.c_int, [_P, _P, _i64, _i64, _i64, _P, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ktf_flat_row_map_rows": (_i64, [_i64, _i64]),
    "ktf_flat_row_map": (C.c_int, [_P, _i64, _i64, _P, _P]),
    "ktf_flat_stats_slots": (_i64, [_i64]),
    "ktf_stats_finalize_flat": (C.c_int, [_P, _i64, _P, _i64, _i64, _i32, _i32, _f32, _P, _i64, _P]),
    "ktf_split_bf16": (C.c_int, [_P, _i64, _i32, _i64, _P, _P, _i64, _P]),
    "ktf_split_bf16_rows": (C.c_int, [_P, _i64, _i64, _i32, _i64, _P, _P, _P, _i64, _P]),
    "ktf_mx_planes": (C.c_int, [_P, _i64, _i64, _i32, _i64, _P, _P, _P, _P, _P, _P]),
    "ktf_tdnn_mx": (C.c_int, [_P, _P, _P, _P, _i64, _i64, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i64, _P]),
    "ktf_tdnn_mx_flat": (C.c_int, [_P, _P, _P, _P, _i64, _i64, _P, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ktf_tdnn_mx_flat_stats": (C.c_int, [_P, _P, _P, _P, _i64, _i64, _P, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ktf_tdnn_mx_stats": (C.c_int, [_P, _P, _P, _P, _i64, _i64, _P, C.POINTER(TdnnDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ktf_stats_finalize": (C.c_int, [_P, _P, _i64, _i64, _i32, _i32, _f32, _P, _i64, _P]),
    "ktf_stats_slots": (_i64, [_i64]),
    "ktf_tdnn_stats_slots": (_i64, [_i64, _i32]),
    "ktf_tdnn_slot_rows": (_i32, [_i32]),
    "ktf_mx_stats_slots": (_i64, [_i64, _i32]),
    "ktf_mx_slot_rows": (_i32, [_i32]),
    "ktf_stats_finalize_slots": (C.c_int, [_P, _i64, _i32, _P, _i64, _i64, _i32, _i32, _f32, _P, _i64, _P]),
    "ktf_affine_act_f32": (C.c_int, [_P, _i64, _i32, _i32, _P, _P, _P, _P]),
    "ktf_activation_f32": (C.c_int, [_P, _i64, _i64, _i32, _i64, _P, _i32, _P, _P, _P]),
    "ktf_convert_pad": (C.c_int, [_P, _i32, _i64, _i32, _i64, _P, _i32, _i64, _P]),
    "ktf_stats_pool": (C.c_int, [_P, _i32, _i64, _i64, _i32, _i64, _P, _i32, _i32, _f32, _P, _i64, _P]),
    "ktf_stats_pool_windowed_f32": (C.c_int, [_P, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _f32, _P, _P]),
    "ktf_xvec_post_f32": (C.c_int, [_P, _i64, _i32, _i32, _P, _P, _P, _P, _P]),
    "ktf_xvec_tail_f32": (C.c_int, [_P, _i64, _P, _i64, _i32, _P, _i64, _i64, _i32, _i32, _f32, _P, _i64, _P, _i32, _P, _P, _P, _i32, _P, _P, _P, _P, _i32, _i32, _P]),
    "ktf_plda_f64": (C.c_int, [_P, _i64, _i32, _P, _P, _P, _i32, _i32, _P, _P, _P]),
    "ktf_plda_f32": (C.c_int, [_P, _i64, _i32, _P, _P, _P, _i32, _i32, _P, _P, _P]),
    "ktf_plda_score_f64": (C.c_int, [_P, _i64, _P, _i64, _i32, _P, _P, _P]),
    "ktf_plda_score_f32": (C.c_int, [_P, _i64, _P, _i64, _i32, _P, _P, _P]),
}

_lib = None


def load():
    """dlopen libktf_hip.so and bind every symbol of the header. Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KtfBackendError(
            f"{LIB_PATH} not found: build it with `make -C kaldi-tflite_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    lib = load()
    buf = C.create_string_buffer(512)
    lib.ktf_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, what):
    """Map the C return code to the reference's exception convention (ValueError for bad arguments)."""
    if rc == 0:
        return
    msg = last_error()
    if rc == -1:
        raise ValueError(msg or what)
    if rc == -3:
        raise NotImplementedError(msg or what)
    raise KtfBackendError(f"{what}: rc={rc}: {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise KtfBackendError("no MI355X / ROCm device visible: kaldi_tflite_amd computes only on the GPU (no CPU fallback)")
    load()


class _Scope(threading.local):
    dev = None          # device index the enclosing launch_scope made current
    stream = None       # ... and its current stream at entry, as a c_void_p


_scope = _Scope()


class launch_scope:
    """One extraction = ~10 launches on one device and one stream: looked up once here instead of once per launch (torch's
    current_stream / device context managers were a quarter of the host time of a batch-1 call). Inside the scope `on_device(d)` is a
    no-op for the scope's device and `stream_ptr()` returns the stream that was current at entry; nested scopes fall back to the
    per-call lookups, and `on_device(another device)` suspends the scope (device and stream) until it exits. Code that switches streams must do so OUTSIDE (as extract_stream does)."""

    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        self.ctx = None
        self.outer = None

    def __enter__(self):
        import torch
        self.outer = (_scope.dev, _scope.stream)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if _scope.dev != idx:
            self.ctx = torch.cuda.device(idx)
            self.ctx.__enter__()
        _scope.dev = idx
        _scope.stream = C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
        return self

    def __exit__(self, *exc):
        _scope.dev, _scope.stream = self.outer
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class _Noop:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NOOP = _Noop()


class _OtherDevice:
    """`on_device(d)` for a device that is not the enclosing launch_scope's: makes it current and hides the scope's cached stream
    for the duration -- that stream belongs to the scope's device, and `stream_ptr()` must hand the call the target device's
    current stream (an invalid handle or a launch on the wrong queue otherwise)."""

    def __init__(self, device):
        import torch
        self.ctx = torch.cuda.device(device)
        self.saved = None

    def __enter__(self):
        self.saved = (_scope.dev, _scope.stream)
        _scope.dev, _scope.stream = None, None
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        _scope.dev, _scope.stream = self.saved
        return False


def on_device(device):
    """`with on_device(t.device):` around a library call: makes the device current (torch.cuda.device), unless an enclosing
    launch_scope already did; for any other device the scope is suspended (its cached stream is not that device's)."""
    import torch
    idx = device.index if isinstance(device, torch.device) else device
    if _scope.dev is not None and (idx is None or idx == _scope.dev):
        return _NOOP
    if _scope.dev is not None:
        return _OtherDevice(device)
    return torch.cuda.device(device)


def stream_ptr():
    if _scope.stream is not None:
        return _scope.stream
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
