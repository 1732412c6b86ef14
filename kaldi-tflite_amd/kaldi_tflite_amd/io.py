"""
ktf.io — readers for Kaldi binary objects (nnet3 raw models, PLDA models, vectors/matrices).

Host-side, one-time (weight loading). Same class names, attributes and return conventions as
kaldi_tflite/lib/io/kaldi/{object_reader,nnet3_reader,plda_reader,array_reader}.py of the
reference; the scanning is done on a memoryview with bytes.find instead of a per-byte loop.
`KaldiIvecExtractorReader` is out of scope.
"""

import re

import numpy as np


class KaldiObjReader:
    """io/kaldi/object_reader.py:23 — cursor over a Kaldi binary file: tokens, basic types, vectors, matrices."""

    def __init__(self, path, binary):
        self.curPos = 0
        self.path = path
        self.binary = binary
        if not self.binary:
            raise NotImplementedError("objects in text format are currently not supported")
        with open(path, "rb") as fd:
            self.data = fd.read()

    # -- raw access
    def readBytes(self, nBytes):
        if self.curPos >= len(self.data):
            return []
        buf = self.data[self.curPos:self.curPos + nBytes]
        self.curPos += len(buf)
        return buf

    def peekBytes(self, nBytes):
        if self.curPos >= len(self.data):
            return []
        return self.data[self.curPos:self.curPos + nBytes]

    def expectLine(self):
        i = self.data.find(b"\n", self.curPos)
        if i < 0:
            raise ValueError("expected new line but did not get any")
        self.curPos = i + 1

    def readLine(self):
        i = self.data.find(b"\n", self.curPos)
        if i < 0:
            raise ValueError("expected new line but did not get any")
        line = self.data[self.curPos:i].decode()
        self.curPos = i + 1
        return line

    def expectToken(self, token, stopTokens=()):
        """Scan forward to `token` (cursor lands one byte past it). If a stop token comes first, leave the cursor
        untouched and return False (object_reader.py:147-196)."""
        tb = token.encode("utf-8")
        hit = self.data.find(tb, self.curPos, max(len(self.data) - 1, 0))
        stop = -1
        for t in stopTokens:
            j = self.data.find(t.encode("utf-8"), self.curPos)
            if j >= 0 and (stop < 0 or j < stop):
                stop = j
        if hit >= 0 and (stop < 0 or hit <= stop):
            self.curPos = hit + len(tb) + 1
            return True
        if stop >= 0:
            return False
        raise ValueError(f"failed to find expected token '{token}")

    def readToken(self):
        i = self.curPos
        while True:
            i = self.data.find(b" ", i)
            if i < 0:
                raise ValueError(f"no whitespace separated token after pos {self.curPos}")
            try:
                token = self.data[self.curPos:i].decode()
                self.curPos = i + 1
                return token
            except UnicodeDecodeError:
                i += 1

    # -- basic types: 1 size byte + little-endian value
    def readBasicType(self, dtype):
        want = np.dtype(dtype).itemsize
        got = int.from_bytes(self.readBytes(1), "little")
        if got != want:
            raise ValueError(f"data type read is specified using {got} bytes, but want to parse {want} bytes")
        buf = self.readBytes(got)
        parsed = np.frombuffer(buf, dtype=dtype)
        if len(parsed) == 0:
            raise ValueError(f"failed to parse any value of type {dtype}")
        return parsed[0]

    def readInt(self):
        return self.readBasicType(np.int32)

    def readFloat(self):
        return self.readBasicType(np.float32)

    def readDouble(self):
        return self.readBasicType(np.float64)

    def readBool(self):
        b = self.readBytes(1)
        if b == b"T":
            return True
        if b == b"F":
            return False
        raise ValueError(f"unexpected format for booleans, expected 'T' or 'F', got {b}")

    # -- containers
    def _dim(self):
        nb = int.from_bytes(self.readBytes(1), "little")
        assert nb == 4
        return int(np.frombuffer(self.readBytes(nb), dtype=np.int32, count=1)[0])

    def readVec(self):
        header = bytes(self.readBytes(3)).decode()
        if header == "FV ":
            size, dt = 4, np.float32
        elif header == "DV ":
            size, dt = 8, np.float64
        else:
            raise ValueError(f"unknown header for vector type '{header.encode()}'")
        n = self._dim()
        if n == 0:
            return np.array([], dtype=dt)
        return np.frombuffer(self.readBytes(n * size), dtype=dt)

    def readMat(self):
        header = bytes(self.readBytes(3)).decode()
        if header.startswith("CM"):
            raise NotImplementedError("can't decode compressed matrix yet")
        elif header == "FM ":
            size, dt = 4, np.float32
        elif header == "DM ":
            size, dt = 8, np.float64
        else:
            raise ValueError(f"unknown header for matrix type '{header}'")
        rows = self._dim()
        cols = self._dim()
        if rows == 0 or cols == 0:
            return np.zeros((rows, cols), dtype=dt)
        return np.frombuffer(self.readBytes(rows * cols * size), dtype=dt).reshape(rows, cols)

    def readPackedMat(self):
        header = bytes(self.readBytes(3)).decode()
        if header == "FP ":
            size, dt = 4, np.float32
        elif header == "DP ":
            size, dt = 8, np.float64
        else:
            raise ValueError(f"unknown header for matrix type '{header}'")
        rows = int(self.readInt())
        if rows == 0:
            return np.zeros((rows, rows), dtype=dt)
        n = (rows + 1) * rows // 2
        sym = np.frombuffer(self.readBytes(n * size), dtype=dt)
        full = np.zeros((rows, rows), dtype=dt)
        il = np.tril_indices(rows)
        full[il] = sym
        full.T[il] = sym
        return full


class KaldiNnet3Reader(KaldiObjReader):
    """io/kaldi/nnet3_reader.py:27 — <Nnet3> raw model: config lines + components with their parameters."""

    def __init__(self, nnet3_path, binary):
        super().__init__(nnet3_path, binary)
        self.config = []
        self.components = []
        self.read()

    def read(self):
        self.expectToken("<Nnet3>")
        line = self.readLine()
        if line.strip() != "":
            raise ValueError("expected model config following <Nnet3> token, got blank line")
        self.readConfigLines()
        self.expectToken("<NumComponents>")
        n = self.readInt()
        assert 0 < n < 100000, f"expected between 1 and 9999 components, got {n}"
        self.components = []
        for _ in range(n):
            self.expectToken("<ComponentName>")
            name = self.readToken()
            ctype = self.readToken()
            comp = {"name": name, "type": ctype}
            comp.update(self.readComponent(ctype))
            self.components.append(comp)
        self.expectToken("</Nnet3>")

    def readConfigLines(self):
        self.config = []
        line = self.readLine().strip()
        while line != "":
            self.config.append(line)
            line = self.readLine().strip()

    def readComponent(self, compType):
        closing = {"</" + compType[1:], "<ComponentName>"}
        data = {}
        for token, fn, key in self.getComponentFormat(compType):
            if self.expectToken(token, closing):
                data[key] = fn()
            else:
                print(f"  - failed to find token {token}")
        return data

    def getComponentFormat(self, compType):
        comp = self.stripTagsAndSuffix(compType, suffix="Component")
        if comp in {"Sigmoid", "Tanh", "RectifiedLinear", "Softmax", "LogSoftmax", "NoOp"}:
            return [("<Dim>", self.readInt, "dim"), ("<ValueAvg>", self.readVec, "value-avg"),
                    ("<DerivAvg>", self.readVec, "deriv-avg"), ("<Count>", self.readDouble, "count"),
                    ("<OderivRms>", self.readVec, "oderiv-rms"), ("<OderivCount>", self.readDouble, "oderiv-count")]
        if comp in {"Affine", "NaturalGradientAffine"}:
            return [("<LinearParams>", self.readMat, "params"), ("<BiasParams>", self.readVec, "bias")]
        if comp == "Linear":
            return [("<Params>", self.readMat, "params")]
        if comp == "BatchNorm":
            return [("<Dim>", self.readInt, "dim"), ("<BlockDim>", self.readInt, "block-dim"),
                    ("<Epsilon>", self.readFloat, "epsilon"), ("<TargetRms>", self.readFloat, "target-rms"),
                    ("<TestMode>", self.readBool, "test-mode"), ("<Count>", self.readDouble, "count"),
                    ("<StatsMean>", self.readVec, "stats-mean"), ("<StatsVar>", self.readVec, "stats-var")]
        if comp in {"StatisticsExtraction", "StatisticsPooling"}:
            return []
        raise ValueError(f"unsupported component type '{compType}'")

    def stripTagsAndSuffix(self, token, suffix=""):
        if token.startswith("<"):
            token = token.lstrip("<")
        if token.endswith("/>"):
            token = token.rstrip("/>")
        if token.endswith(">"):
            token = token.rstrip(">")
        if suffix and token.endswith(suffix):
            token = token[:len(token) - len(suffix)]
        return token

    def getComponent(self, name):
        return [c for c in self.components if c.get("name") is not None and re.match(f"{name}", c["name"])]

    def getWeights(self, name):
        comps = self.getComponent(name)
        if len(comps) == 0:
            raise KeyError(f"no components with name matching '{name}'")
        weights = []
        for c in comps:
            t = c["type"]
            if t == "<NaturalGradientAffineComponent>":
                weights.extend([c["params"], c["bias"]])
            elif t == "<BatchNormComponent>":
                weights.extend([c["target-rms"], c["stats-mean"], c["stats-var"]])
        return weights


class KaldiPldaReader(KaldiObjReader):
    """io/kaldi/plda_reader.py:22 — <Plda> mean, transform, psi."""

    def __init__(self, plda_path, binary):
        super().__init__(plda_path, binary)
        self.mean = self.transformMat = self.psi = None
        self.read()

    def read(self):
        self.expectToken("<Plda>")
        self.mean = self.readVec()
        self.transformMat = self.readMat()
        self.psi = self.readVec()
        self.expectToken("</Plda>")


def ReadKaldiArray(path, binary, dtype=np.float32):
    """io/kaldi/array_reader.py:24 — one vector or matrix from a binary (\\0B + FV/DV/FM/DM) or text ([ ... ]) file."""
    if binary:
        r = KaldiObjReader(path, True)
        r.readBytes(2)
        kind = bytes(r.peekBytes(2)).decode()
        if kind in ["FM", "DM", "CM"]:
            return r.readMat()
        if kind in ["FV", "DV"]:
            return r.readVec()
        raise ValueError(f"binary file contains unexpected header bytes, {kind}, expected 'FV', 'DV', 'FM', 'DM' or 'CM'")

    if dtype in [np.float32, np.float64]:
        conv = float
    elif dtype in [np.int16, np.int32, np.int64]:
        conv = int
    else:
        raise ValueError(f"unsupported data type: {dtype}")
    mat = []
    with open(path, "r") as f:
        for line in f:
            toks = line.strip().split()
            if "[" in toks and "]" in toks:
                return np.array([conv(t) for t in toks[1:-1]], dtype=dtype)
            if "[" in toks:
                if len(toks) > 1:
                    mat.append([conv(t) for t in toks[1:]])
                continue
            if "]" in toks:
                if len(toks) > 1:
                    mat.append([conv(t) for t in toks[:-1]])
                return np.array(mat, dtype=dtype)
            mat.append([conv(t) for t in toks])
    raise ValueError("reached end of file without finding closing bracket for matrix")
