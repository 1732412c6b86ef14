"""
Host side of KTF_GEMM_F16MX (include/ktf_hip.h, csrc/tdnn_mx.hip): the OCP-MX element codecs (e2m1 "fp4", e2m3 "fp6", E8M0
block scales), the weight images the kernel streams, and the four-plane activation container.

    y = x_h w_h + x_l4 w_4 + x_4 w_l6        x_h, w_h half;  *_4 e2m1 images;  w_l6 e2m3 image of w - w_h

One power-of-two scale per 32 consecutive K elements. The weights are encoded here once per model (NumPy, float64 in);
activations are encoded on the device (ktf_mx_planes, the GEMM epilogue) with the same rules, so `decode_*` below is what
the tests use to read device planes back.
"""

import numpy as np
import torch

FP4_MAX, FP6_MAX = 6.0, 7.5
WQ_BLOCK = 49152                      # bytes of one (N-tile, super-step) block of the MX weight planes (256-row kernel)
WQ_BLOCK_LOADER = 45056               # ... of the loader-wave kernel's images (two 22 KiB halves)


# ------------------------------------------------------------------------------------------------ element codecs
def scale_bytes(block_max, fmt):
    """E8M0 byte per block: 2^(byte - 127) puts the block maximum into the element format's top binade [4, 8), one binade
    lower when the maximum would round past the largest element (e2m1: >= 7 -> 8 > 6; e2m3: >= 7.75 -> 8 > 7.5). Zero blocks
    and underflow get byte 1."""
    thr = {"e2m1": 1.75, "e2m3": 1.9375}[fmt]
    m = np.asarray(block_max, np.float64)
    mant, ex = np.frexp(m)                                  # m = mant * 2^ex, mant in [0.5, 1)
    e = ex - 1 - 2 + (2.0 * mant >= thr)
    return np.where(m > 0, np.clip(e + 127, 1, 254), 1).astype(np.uint8)


def _scaled(v, sbytes):
    return np.asarray(v, np.float64) / np.exp2(sbytes.astype(np.float64) - 127.0)[..., None]


def encode_e2m1(v, sbytes):
    """(..., 32) values, (...,) scale bytes -> (..., 32) codes 0..15 (sign bit 8; magnitudes 0 .5 1 1.5 2 3 4 6), round to
    nearest even, saturating."""
    u = _scaled(v, sbytes)
    a = np.abs(u)
    k = np.where(a < 2.0, np.rint(a * 2.0), np.where(a < 4.0, np.rint(a) + 2.0, np.rint(a * 0.5) + 4.0))
    k = np.minimum(k, 7.0).astype(np.uint8)
    return (k | np.where(np.signbit(u), 8, 0).astype(np.uint8)).astype(np.uint8)


def decode_e2m1(codes, sbytes):
    mag = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])[codes & 7]
    return np.where(codes & 8, -mag, mag) * np.exp2(sbytes.astype(np.float64) - 127.0)[..., None]


def encode_e2m3(v, sbytes):
    """(..., 32) values -> codes 0..63 (sign bit 32, 2 exponent bits, 3 mantissa bits; subnormal step 1/8, largest 7.5)."""
    u = _scaled(v, sbytes)
    a = np.abs(u)
    e = np.clip(np.floor(np.log2(np.maximum(a, 1.0))), 0, 2)
    k = np.minimum(8.0 * e + np.rint(a / np.exp2(e) * 8.0), 31.0).astype(np.uint8)
    return (k | np.where(np.signbit(u), 32, 0).astype(np.uint8)).astype(np.uint8)


def decode_e2m3(codes, sbytes):
    k = (codes & 31).astype(np.int64)
    e, m = k >> 3, k & 7
    mag = np.where(e == 0, m / 8.0, (1.0 + m / 8.0) * np.exp2(e - 1.0))
    return np.where(codes & 32, -mag, mag) * np.exp2(sbytes.astype(np.float64) - 127.0)[..., None]


def pack4(codes):
    """(..., 32) e2m1 codes -> (..., 16) bytes, element e in nibble e (low nibble first)."""
    c = codes.astype(np.uint8)
    return (c[..., 0::2] | (c[..., 1::2] << 4)).astype(np.uint8)


def unpack4(b):
    b = np.asarray(b, np.uint8)
    out = np.empty(b.shape[:-1] + (b.shape[-1] * 2,), np.uint8)
    out[..., 0::2] = b & 15
    out[..., 1::2] = b >> 4
    return out


def pack6(codes):
    """(..., 32) e2m3 codes -> (..., 24) bytes, element e at bit 6 e of the little-endian 192-bit string."""
    bits = (codes.astype(np.uint8)[..., None] >> np.arange(6, dtype=np.uint8)) & 1
    return np.packbits(bits.reshape(codes.shape[:-1] + (192,)), axis=-1, bitorder="little")


# ------------------------------------------------------------------------------------------------ activation planes
def encode_activations(x):
    """float array (..., D) with D % 32 == 0 -> (x_h half (..., D), l4 codes (..., D), x4 codes (..., D), scale bytes l4, x4
    (..., D / 32)): the arithmetic of mx_encode32 in csrc/tdnn_mx.hip, for tests."""
    x = np.clip(np.asarray(x, np.float32), -65504.0, 65504.0)
    xh = x.astype(np.float16)
    lo = (x - xh.astype(np.float32)).astype(np.float64)
    blk = x.shape[:-1] + (x.shape[-1] // 32, 32)
    hb, lb = xh.astype(np.float64).reshape(blk), lo.reshape(blk)
    sh, sl = scale_bytes(np.abs(hb).max(-1), "e2m1"), scale_bytes(np.abs(lb).max(-1), "e2m1")
    return xh, encode_e2m1(lb, sl).reshape(x.shape), encode_e2m1(hb, sh).reshape(x.shape), sl, sh


class Planes:
    """The four chunk-major device planes of one (B, T, D) activation (csrc/tdnn_mx.hip): xh (B, nch, T, 32) half,
    xl4 / x4 (B, nch, T, 16) uint8, xs (B, nch, T) int32."""

    def __init__(self, xh, xl4, x4, xs, D):
        self.xh, self.xl4, self.x4, self.xs, self.D = xh, xl4, x4, xs, D

    @property
    def shape(self):
        return (self.xh.shape[0], self.xh.shape[2], self.D)

    @property
    def device(self):
        return self.xh.device

    @staticmethod
    def buffers(get, role, B, T, D, device):
        """Planes over workspace views: `get(role, shape, dtype, device, padded=False)`."""
        nch = (D + 31) // 32
        return Planes(get(role + "_h", (B, nch, T, 32), torch.float16, device, padded=False),
                      get(role + "_l4", (B, nch, T, 16), torch.uint8, device, padded=False),
                      get(role + "_4", (B, nch, T, 16), torch.uint8, device, padded=False),
                      get(role + "_s", (B, nch, T), torch.int32, device, padded=False), D)

    @staticmethod
    def empty(B, T, D, device):
        nch = (D + 31) // 32
        return Planes(torch.zeros((B, nch, T, 32), dtype=torch.float16, device=device),
                      torch.zeros((B, nch, T, 16), dtype=torch.uint8, device=device),
                      torch.zeros((B, nch, T, 16), dtype=torch.uint8, device=device),
                      torch.zeros((B, nch, T), dtype=torch.int32, device=device), D)

    def decode(self):
        """-> (x_h, x_l4, x_4) float64 arrays (B, T, nch * 32): what the three products of the kernel see."""
        xh = self.xh.cpu().numpy().astype(np.float64)
        s = self.xs.cpu().numpy().astype(np.uint32)
        l4 = decode_e2m1(unpack4(self.xl4.cpu().numpy()), (s & 255).astype(np.uint8))
        h4 = decode_e2m1(unpack4(self.x4.cpu().numpy()), ((s >> 8) & 255).astype(np.uint8))
        B, nch, T, _ = xh.shape
        to_rows = lambda a: a.transpose(0, 2, 1, 3).reshape(B, T, nch * 32)
        return to_rows(xh), to_rows(l4), to_rows(h4)


# ------------------------------------------------------------------------------------------------ weight images
def weight_images(Wk, permuted=False):
    """Wk: float64 (Up, nk, 32) -- units padded to a multiple of 256, K-steps in the KTF_TDNN_K_INTERLEAVED order -> (wh bytes,
    wq bytes, decoded (w_h, w_4, w_l6) float64 (Up, nkp, 32)) in the layouts of include/ktf_hip.h (ktf_tdnn_mx). `permuted` (the
    experiments under tools/mx/experiments/ that run the MFMAs with the weights as the A operand): the image rows of every 256-unit
    tile hold the units in loader_unit_order(), so that a lane owns eight consecutive units of a frame. The decoded operands stay
    in unit order."""
    Up, nk, _ = Wk.shape
    nkp = (nk + 3) // 4 * 4
    W = np.zeros((Up, nkp, 32), np.float64)
    W[:, :nk] = Wk
    if permuted:
        wh, wq, _ = weight_images(W.reshape(Up // 256, 256, nkp, 32)[:, loader_unit_order()].reshape(Up, nkp, 32), permuted=False)
        return wh, wq, weight_images(W, permuted=False)[2]
    Wh = W.astype(np.float16)
    Wl = W - Wh.astype(np.float64)
    s4, s6 = scale_bytes(np.abs(W).max(-1), "e2m1"), scale_bytes(np.abs(Wl).max(-1), "e2m3")
    c4, c6 = encode_e2m1(W, s4), encode_e2m3(Wl, s6)
    nt, nss = Up // 256, nkp // 4
    # wh: per (N-tile, K-step) the 256 x 64-byte LDS image: row r keeps its four 16-byte chunks at positions chunk ^ ((4 - (r >> 2)) & 3)
    r = np.arange(256)
    src = np.arange(4)[None, :] ^ ((4 - ((r >> 2) & 3)) & 3)[:, None]                  # [row, position] -> chunk
    W5 = Wh.reshape(nt, 256, nkp, 4, 8)
    wh = np.ascontiguousarray(W5[:, r[:, None], :, src, :].transpose(2, 3, 0, 1, 4))     # (nt, nkp, 256, 4 positions, 8)
    # wq: per (N-tile, super-step): [kb][col] records
    p4 = pack4(c4).reshape(nt, 256, nss, 4, 16).transpose(0, 2, 3, 1, 4)               # (nt, nss, kb, col, 16)
    p6 = pack6(c6).reshape(nt, 256, nss, 4, 24).transpose(0, 2, 3, 1, 4)               # (nt, nss, kb, col, 24)
    sc = (s4.astype(np.uint32) | (s6.astype(np.uint32) << 8)).reshape(nt, 256, nss, 4).transpose(0, 2, 3, 1)
    wq = np.zeros((nt, nss, WQ_BLOCK), np.uint8)
    wq[:, :, 0:16384] = p4.reshape(nt, nss, 16384)
    wq[:, :, 16384:32768] = np.ascontiguousarray(p6[..., :16]).reshape(nt, nss, 16384)
    wq[:, :, 32768:40960] = np.ascontiguousarray(p6[..., 16:]).reshape(nt, nss, 8192)
    wq[:, :, 40960:45056] = np.ascontiguousarray(sc).view(np.uint8).reshape(nt, nss, 4096)
    dec = (Wh.astype(np.float64), decode_e2m1(c4, s4), decode_e2m3(c6, s6))
    return wh.view(np.uint8).reshape(-1), wq.reshape(-1), dec


def loader_unit_order():
    """(256,) unit (inside a 256-unit tile) held by column `m` of unit block `cb` (index cb * 16 + m) of the loader kernel's weight
    images: inside every 32-unit chunk the order is permuted so that, with the weights as the A operand of the MFMAs, the four lanes
    of a frame own units 0-7, 8-15, 16-23, 24-31 of the chunk (csrc/tdnn_mxl.hip, xl_unit)."""
    cb, m = np.divmod(np.arange(256), 16)
    return (cb >> 1) * 32 + (m >> 2) * 8 + (cb & 1) * 4 + (m & 3)


def weight_images_loader(Wk):
    """As weight_images, in the layouts of the loader-wave kernel (csrc/tdnn_mxl.hip; include/ktf_hip.h, KTF_TDNN_MX_LOADER): every
    operand fragment is 64 lanes x 16 (8, 4) consecutive bytes, lane = 16 * (K quarter / K block) + column.
      wh: per (N-tile, K-step) 16 fragments (unit blocks cb) x 1 KiB: lane (q, m) holds halves 8 q .. 8 q + 7 of unit order[16 cb + m]
      wq: per (N-tile, super-step) two halves of 22 KiB, half h = unit blocks cb with (cb >> 1) & 1 == h, in the order
          cbh = 2 (cb >> 2) + (cb & 1):  e2m1 codes 8 x 1 KiB | e2m3 bits 0..127 8 x 1 KiB | e2m3 bits 128..191 8 x 512 B |
          scale words 8 x 256 B; lane (kb, m) = K block kb of the super-step, unit order[16 cb + m].
    The decoded operands are the same numbers as weight_images' (the codecs do not depend on the layout)."""
    Up, nk, _ = Wk.shape
    nkp = (nk + 3) // 4 * 4
    W = np.zeros((Up, nkp, 32), np.float64)
    W[:, :nk] = Wk
    Wh = W.astype(np.float16)
    Wl = W - Wh.astype(np.float64)
    s4, s6 = scale_bytes(np.abs(W).max(-1), "e2m1"), scale_bytes(np.abs(Wl).max(-1), "e2m3")
    c4, c6 = encode_e2m1(W, s4), encode_e2m3(Wl, s6)
    nt, nss = Up // 256, nkp // 4
    order = loader_unit_order()                                                       # image column -> unit of the tile
    tile = lambda a: a.reshape((nt, 256) + a.shape[1:])[:, order]                     # (nt, 256 image columns, ...)
    # wh: (nt, ks, cb, q, m, 8)
    wh = np.ascontiguousarray(tile(Wh).reshape(nt, 16, 16, nkp, 4, 8).transpose(0, 3, 1, 4, 2, 5))
    # wq halves: image columns -> (cb, m); half h = (cb >> 1) & 1; inside a half cbh = 2 (cb >> 2) + (cb & 1)
    cb = np.arange(16)
    sel = [np.array(sorted(cb[((cb >> 1) & 1) == h], key=lambda c: 2 * (c >> 2) + (c & 1))) for h in (0, 1)]
    p4 = tile(pack4(c4)).reshape(nt, 16, 16, nss, 4, 16)                              # (nt, cb, m, ss, kb, 16)
    p6 = tile(pack6(c6)).reshape(nt, 16, 16, nss, 4, 24)
    sc = tile((s4.astype(np.uint32) | (s6.astype(np.uint32) << 8))).reshape(nt, 16, 16, nss, 4)
    wq = np.zeros((nt, nss, 2, WQ_BLOCK_LOADER // 2), np.uint8)
    for h in (0, 1):
        a4 = p4[:, sel[h]].transpose(0, 3, 1, 4, 2, 5)                                # (nt, ss, cbh, kb, m, 16)
        a6 = p6[:, sel[h]].transpose(0, 3, 1, 4, 2, 5)
        asc = sc[:, sel[h]].transpose(0, 3, 1, 4, 2)                                  # (nt, ss, cbh, kb, m)
        wq[:, :, h, 0:8192] = np.ascontiguousarray(a4).reshape(nt, nss, 8192)
        wq[:, :, h, 8192:16384] = np.ascontiguousarray(a6[..., :16]).reshape(nt, nss, 8192)
        wq[:, :, h, 16384:20480] = np.ascontiguousarray(a6[..., 16:]).reshape(nt, nss, 4096)
        wq[:, :, h, 20480:22528] = np.ascontiguousarray(asc).view(np.uint8).reshape(nt, nss, 2048)
    dec = (Wh.astype(np.float64), decode_e2m1(c4, s4), decode_e2m3(c6, s6))
    return wh.view(np.uint8).reshape(-1), wq.reshape(-1), dec
