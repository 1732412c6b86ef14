"""
ktf.io — Kaldi binary object readers (nnet3 raw models, PLDA models, vectors / matrices).

Host-side, one-time (weight loading). The public names and the data they return follow the
reference (kaldi_tflite/lib/io/kaldi/{object_reader,nnet3_reader,plda_reader,array_reader}.py:
`KaldiObjReader`, `KaldiNnet3Reader.{config,components,getWeights}`,
`KaldiPldaReader.{mean,transformMat,psi}`, `ReadKaldiArray`); the parser is not the reference's
token search. Kaldi's binary stream is self-delimiting --

    "<Tag> "                      a tag, followed by one of
    0x04 + 4 bytes | 0x08 + 8 bytes      a basic type (size byte + little-endian value)
    "T" | "F"                     a bool
    "FV "/"DV " + dim + data      a vector      (dim = 0x04 + int32)
    "FM "/"DM " + rows + cols + data     a matrix
    "FP "/"DP " + rows + data     a packed symmetric matrix
    text + " "                    a word (component name, ...)
    "<"                           nothing: the next tag

-- so an nnet3 file is decoded in ONE forward pass into (tag, payload) events (`iter_fields`), and a
component is the run of events between two <ComponentName> tags. `KaldiIvecExtractorReader` is out of
scope.
"""

import re
import struct

import numpy as np

_I32 = struct.Struct("<i")
_F32 = struct.Struct("<f")
_F64 = struct.Struct("<d")
_CONTAINERS = {b"FV ": ("vec", np.float32), b"DV ": ("vec", np.float64), b"FM ": ("mat", np.float32),
               b"DM ": ("mat", np.float64), b"FP ": ("packed", np.float32), b"DP ": ("packed", np.float64)}

# 4-byte basic types are ambiguous on the wire (int32 or float32): the tags whose value is an integer
_INT_TAGS = {"<Dim>", "<BlockDim>", "<NumComponents>", "<RankIn>", "<RankOut>", "<UpdatePeriod>", "<InputDim>",
             "<OutputDim>", "<NumRepeats>", "<NumBlocks>"}
# tag -> key under which KaldiNnet3Reader stores the payload in a component dict (the reference's key names,
# nnet3_reader.py:189-225; other tags of a component are decoded and dropped)
_FIELD_KEYS = {
    "<Dim>": "dim", "<ValueAvg>": "value-avg", "<DerivAvg>": "deriv-avg", "<Count>": "count",
    "<OderivRms>": "oderiv-rms", "<OderivCount>": "oderiv-count",
    "<LinearParams>": "params", "<BiasParams>": "bias", "<Params>": "params",
    "<BlockDim>": "block-dim", "<Epsilon>": "epsilon", "<TargetRms>": "target-rms", "<TestMode>": "test-mode",
    "<StatsMean>": "stats-mean", "<StatsVar>": "stats-var",
}
# component kinds (type tag without "<", "Component>") whose fields are kept; anything else is rejected like the reference
_KINDS = {
    "Sigmoid": ("dim", "value-avg", "deriv-avg", "count", "oderiv-rms", "oderiv-count"),
    "Affine": ("params", "bias"),
    "Linear": ("params",),
    "BatchNorm": ("dim", "block-dim", "epsilon", "target-rms", "test-mode", "count", "stats-mean", "stats-var"),
    "StatisticsExtraction": (),
}
for _alias, _of in (("Tanh", "Sigmoid"), ("RectifiedLinear", "Sigmoid"), ("Softmax", "Sigmoid"), ("LogSoftmax", "Sigmoid"),
                    ("NoOp", "Sigmoid"), ("NaturalGradientAffine", "Affine"), ("StatisticsPooling", "StatisticsExtraction")):
    _KINDS[_alias] = _KINDS[_of]


class KaldiObjReader:
    """io/kaldi/object_reader.py:23 — a cursor (`curPos`) over the bytes of one Kaldi binary file with typed reads.
    Text-mode objects are not supported (NotImplementedError), as in the reference (:46-47)."""

    def __init__(self, path, binary):
        if not binary:
            raise NotImplementedError("objects in text format are currently not supported")
        self.path, self.binary, self.curPos = path, binary, 0
        with open(path, "rb") as f:
            self.data = f.read()

    # ---- raw bytes
    def _take(self, n):
        lo = self.curPos
        if lo + n > len(self.data):
            raise ValueError(f"{self.path}: truncated ({n} bytes wanted at offset {lo}, file has {len(self.data)})")
        self.curPos = lo + n
        return lo

    def readBytes(self, nBytes):
        lo = self.curPos
        self.curPos = min(len(self.data), lo + nBytes)
        return self.data[lo:self.curPos]

    def peekBytes(self, nBytes):
        return self.data[self.curPos:self.curPos + nBytes]

    def readLine(self):
        end = self.data.find(b"\n", self.curPos)
        if end < 0:
            raise ValueError("expected new line but did not get any")
        line = self.data[self.curPos:end].decode("utf-8", "replace")
        self.curPos = end + 1
        return line

    def expectLine(self):
        self.readLine()

    # ---- words and tags
    def readToken(self):
        end = self.data.find(b" ", self.curPos)
        if end < 0:
            raise ValueError(f"no whitespace separated token after pos {self.curPos}")
        word = self.data[self.curPos:end].decode("utf-8", "replace")
        self.curPos = end + 1
        return word

    def expectToken(self, token, stopTokens=()):
        """Moves the cursor just past the next occurrence of `token` (and the separator after it) and returns True; if
        one of `stopTokens` occurs earlier the cursor stays and the result is False (object_reader.py:147-196)."""
        want = token.encode()
        at = self.data.find(want, self.curPos)
        stops = [p for p in (self.data.find(s.encode(), self.curPos) for s in stopTokens) if p >= 0]
        if at >= 0 and not any(p < at for p in stops):
            self.curPos = at + len(want) + 1
            return True
        if stops:
            return False
        raise ValueError(f"failed to find expected token '{token}")

    # ---- basic types: one size byte, then the little-endian value
    def _basic(self, st):
        lo = self._take(1 + st.size)
        if self.data[lo] != st.size:
            raise ValueError(f"data type read is specified using {self.data[lo]} bytes, but want to parse {st.size} bytes")
        return st.unpack_from(self.data, lo + 1)[0]

    def readInt(self):
        return np.int32(self._basic(_I32))

    def readFloat(self):
        return np.float32(self._basic(_F32))

    def readDouble(self):
        return np.float64(self._basic(_F64))

    def readBasicType(self, dtype):
        return {4: self.readInt if np.issubdtype(dtype, np.integer) else self.readFloat, 8: self.readDouble}[np.dtype(dtype).itemsize]()

    def readBool(self):
        c = self.data[self._take(1):self.curPos]
        if c not in (b"T", b"F"):
            raise ValueError(f"unexpected format for booleans, expected 'T' or 'F', got {c}")
        return c == b"T"

    # ---- containers
    def _container(self, *kinds):
        head = bytes(self.peekBytes(3))
        if head[:2] == b"CM":
            raise NotImplementedError("can't decode compressed matrix yet")
        spec = _CONTAINERS.get(head)
        if spec is None or spec[0] not in kinds:
            raise ValueError(f"unknown header for {'/'.join(kinds)} type '{head}'")
        self.curPos += 3
        return spec[1]

    def _array(self, dtype, count):
        lo = self._take(count * np.dtype(dtype).itemsize)
        return np.frombuffer(self.data, dtype=dtype, count=count, offset=lo)

    def readVec(self):
        dtype = self._container("vec")
        return self._array(dtype, int(self.readInt()))

    def readMat(self):
        dtype = self._container("mat")
        rows, cols = int(self.readInt()), int(self.readInt())
        return self._array(dtype, rows * cols).reshape(rows, cols)

    def readPackedMat(self):
        """lower triangle, row by row -> full symmetric matrix (object_reader.py:434-483)."""
        dtype = self._container("packed")
        n = int(self.readInt())
        tri = self._array(dtype, n * (n + 1) // 2)
        full = np.zeros((n, n), dtype=dtype)
        r, c = np.tril_indices(n)
        full[r, c] = tri
        full[c, r] = tri
        return full

    # ---- one-pass event decoding
    def iter_fields(self, end_tag=None):
        """Yields (tag, payload) from the cursor on: payload is None (tag only), a word, a scalar, a bool or an array.
        Stops after `end_tag` (yielded with payload None) or at the end of the data."""
        data, n = self.data, len(self.data)
        while self.curPos < n:
            if data[self.curPos] in b" \n":
                self.curPos += 1
                continue
            if data[self.curPos] != 0x3C:           # not "<": a further value of the previous tag
                yield None, self._payload(None)
                continue
            tag = self.readToken()
            if tag == end_tag:
                yield tag, None
                return
            yield tag, self._payload(tag)

    def _payload(self, tag):
        head = bytes(self.peekBytes(3))
        if not head or head[:1] == b"<":
            return None
        if head[0] == 4:
            return self.readInt() if tag in _INT_TAGS else self.readFloat()
        if head[0] == 8:
            return self.readDouble()
        if head in _CONTAINERS:
            kind = _CONTAINERS[head][0]
            return {"vec": self.readVec, "mat": self.readMat, "packed": self.readPackedMat}[kind]()
        if head[:2] == b"CM":
            raise NotImplementedError("can't decode compressed matrix yet")
        if head[:1] in (b"T", b"F") and (len(head) == 1 or head[1:2] in (b"<", b" ")):
            return self.readBool()
        return self.readToken()


class KaldiNnet3Reader(KaldiObjReader):
    """io/kaldi/nnet3_reader.py:27 — `config`: the lines of the <Nnet3> header; `components`: one dict per component
    ({"name", "type", + the fields of nnet3_reader.py:189-225 that are present}) in file order."""

    def __init__(self, nnet3_path, binary):
        super().__init__(nnet3_path, binary)
        self.config, self.components = [], []
        self.read()

    def read(self):
        self.expectToken("<Nnet3>")
        if self.readLine().strip():
            raise ValueError("expected model config following <Nnet3> token, got blank line")
        self.config = []
        while True:
            line = self.readLine().strip()
            if not line:
                break
            self.config.append(line)
        self.components, cur, declared, closed = [], None, None, False
        for tag, val in self.iter_fields(end_tag="</Nnet3>"):
            if tag is None:
                continue
            if tag == "<NumComponents>":
                declared = int(val)
                assert 0 < declared < 100000, f"expected between 1 and 9999 components, got {declared}"
            elif tag == "<ComponentName>":
                cur = {"name": val, "type": None}
                self.components.append(cur)
            elif tag == "</Nnet3>":
                closed = True
            elif cur is not None and cur["type"] is None:
                cur["type"] = tag
                cur["_keep"] = _KINDS.get(self.stripTagsAndSuffix(tag, "Component"))
                if cur["_keep"] is None:
                    raise ValueError(f"unsupported component type '{tag}'")
                if val is not None:          # the type tag is followed directly by the first field's tag, never by data
                    raise ValueError(f"unexpected data after component type {tag}")
            elif cur is not None:
                key = _FIELD_KEYS.get(tag)
                if key in cur["_keep"] and key not in cur:
                    cur[key] = val
        if declared is None:
            raise ValueError("failed to find expected token '<NumComponents>")
        if not closed:
            raise ValueError("failed to find expected token '</Nnet3>")
        if len(self.components) != declared:
            raise ValueError(f"<NumComponents> says {declared}, file holds {len(self.components)}")
        for c in self.components:
            for key in c.pop("_keep"):
                if key not in c:
                    print(f"  - component {c['name']}: no field '{key}'")

    @staticmethod
    def stripTagsAndSuffix(token, suffix=""):
        """'<FooComponent>' -> 'Foo'."""
        core = token.strip("<>/")
        return core[:-len(suffix)] if suffix and core.endswith(suffix) else core

    def getComponent(self, name):
        """components whose name matches the regular expression `name` from its start (sequential.py:136-138 passes the
        layer name)."""
        pat = re.compile(name)
        return [c for c in self.components if pat.match(c["name"])]

    def getWeights(self, name):
        """[W (units, K*D), b] of an affine component, [target-rms, stats-mean, stats-var] of a batch-norm component,
        concatenated over every component matching `name`; KeyError when none does (nnet3_reader.py:316-324)."""
        found = self.getComponent(name)
        if not found:
            raise KeyError(f"no components with name matching '{name}'")
        order = {"NaturalGradientAffine": ("params", "bias"), "BatchNorm": ("target-rms", "stats-mean", "stats-var")}
        out = []
        for c in found:
            out += [c[k] for k in order.get(self.stripTagsAndSuffix(c["type"], "Component"), ())]
        return out


class KaldiPldaReader(KaldiObjReader):
    """io/kaldi/plda_reader.py:22 — <Plda> mean (dim), transformMat (dim, dim), psi (dim); float64 in Kaldi's files."""

    def __init__(self, plda_path, binary):
        super().__init__(plda_path, binary)
        self.read()

    def read(self):
        self.expectToken("<Plda>")
        self.mean, self.transformMat, self.psi = self.readVec(), self.readMat(), self.readVec()
        self.expectToken("</Plda>")


def _text_array(path, dtype):
    """Kaldi text form: ' [ v v v ]' on one line is a vector; '[' ... rows ... ']' over several lines a matrix."""
    if dtype not in (np.float32, np.float64, np.int16, np.int32, np.int64):
        raise ValueError(f"unsupported data type: {dtype}")
    parse = np.float64 if np.issubdtype(dtype, np.floating) else np.int64
    with open(path, "r") as f:
        text = f.read()
    lo, hi = text.find("["), text.find("]")
    if lo < 0 or hi < lo:
        raise ValueError("reached end of file without finding closing bracket for matrix")
    body = text[lo + 1:hi]
    if "\n" not in body:
        return np.array(body.split(), dtype=parse).astype(dtype)
    rows = [np.array(r.split(), dtype=parse) for r in body.split("\n") if r.strip()]
    return np.array(rows).astype(dtype) if rows else np.zeros((0, 0), dtype=dtype)


def ReadKaldiArray(path, binary, dtype=np.float32):
    """io/kaldi/array_reader.py:24 — one vector or matrix from a binary ("\\0B" + FV/DV/FM/DM) or text ([ ... ]) file."""
    if not binary:
        return _text_array(path, dtype)
    r = KaldiObjReader(path, True)
    r.curPos = 2                                   # "\0B"
    kind = bytes(r.peekBytes(2))
    if kind in (b"FM", b"DM", b"CM"):
        return r.readMat()
    if kind in (b"FV", b"DV"):
        return r.readVec()
    raise ValueError(f"binary file contains unexpected header bytes, {kind.decode(errors='replace')}, expected 'FV', 'DV', 'FM', 'DM' or 'CM'")
