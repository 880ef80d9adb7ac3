"""Minimal GGUF v3 writer/reader (numpy only).

Host-side tooling: the reference converts checkpoints with the third-party `gguf` package
(scripts/convert.py:25,46-98), which is not installed here, and ships no .gguf file. This
module writes files with the same on-disk contract so the C++ loader (csrc/gguf.cpp) can be
exercised, and reads them back independently of that loader for cross-checks.

Format (little endian): "GGUF", u32 version=3, u64 n_tensors, u64 n_kv; KV = string key,
u32 type, value; tensor info = string name, u32 n_dims, u64 ne[n_dims] (ne[0] fastest),
u32 ggml_type, u64 offset; data section aligned to general.alignment (default 32).
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

MAGIC = b"GGUF"
VERSION = 3
ALIGNMENT = 32

# gguf value types
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR_FMT = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f",
               T_BOOL: "<?", T_U64: "<Q", T_I64: "<q", T_F64: "<d"}

# ggml tensor types the reference uses (tests/workbench.py:14-19)
GGML_F32, GGML_F16, GGML_I32 = 0, 1, 26
_NP2GGML = {np.dtype(np.float32): GGML_F32, np.dtype(np.float16): GGML_F16, np.dtype(np.int32): GGML_I32}
_GGML2NP = {v: k for k, v in _NP2GGML.items()}


def _pad(n: int, a: int = ALIGNMENT) -> int:
    return (n + a - 1) // a * a


def _wstr(s: str) -> bytes:
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


class GGUFWriter:
    def __init__(self, path: str | Path, arch: str):
        self.path = Path(path)
        self.kv: list[tuple[str, int, object]] = []
        self.tensors: list[tuple[str, np.ndarray]] = []
        self.add_string("general.architecture", arch)

    def add_string(self, key: str, val: str):
        self.kv.append((key, T_STR, val))

    def add_int32(self, key: str, val: int):
        self.kv.append((key, T_I32, int(val)))

    def add_uint32(self, key: str, val: int):
        self.kv.append((key, T_U32, int(val)))

    def add_float32(self, key: str, val: float):
        self.kv.append((key, T_F32, float(val)))

    def add_array_i32(self, key: str, vals):
        self.kv.append((key, T_ARR, (T_I32, [int(v) for v in vals])))

    def add_tensor(self, name: str, arr: np.ndarray):
        """arr is in torch axis order (slowest first); stored with ne reversed."""
        if len(name) >= 64:
            raise ValueError(f"tensor name too long ({len(name)}): {name}")
        a = np.ascontiguousarray(arr)
        if a.dtype not in _NP2GGML:
            raise TypeError(f"unsupported dtype {a.dtype}")
        self.tensors.append((name, a))

    def write(self):
        out = bytearray()
        out += MAGIC + struct.pack("<IQQ", VERSION, len(self.tensors), len(self.kv))
        for key, typ, val in self.kv:
            out += _wstr(key) + struct.pack("<I", typ)
            if typ == T_STR:
                out += _wstr(val)
            elif typ == T_ARR:
                et, vals = val
                out += struct.pack("<IQ", et, len(vals))
                for v in vals:
                    out += struct.pack(_SCALAR_FMT[et], v)
            else:
                out += struct.pack(_SCALAR_FMT[typ], val)
        offset = 0
        offsets = []
        for name, a in self.tensors:
            out += _wstr(name)
            shape = list(a.shape)[::-1] or [1]
            out += struct.pack("<I", len(shape))
            for d in shape:
                out += struct.pack("<Q", d)
            out += struct.pack("<IQ", _NP2GGML[a.dtype], offset)
            offsets.append(offset)
            offset = _pad(offset + a.nbytes)
        out += b"\0" * (_pad(len(out)) - len(out))
        with open(self.path, "wb") as f:
            f.write(out)
            pos = 0
            for (name, a), off in zip(self.tensors, offsets):
                if off > pos:
                    f.write(b"\0" * (off - pos))
                    pos = off
                f.write(a.tobytes())
                pos += a.nbytes
            f.write(b"\0" * (_pad(pos) - pos))


class GGUFFile:
    """Reads a GGUF file: .kv dict, .tensors dict name -> numpy array (torch axis order)."""

    def __init__(self, path: str | Path):
        buf = Path(path).read_bytes()
        self._b = buf
        self._p = 0
        if self._take(4) != MAGIC:
            raise ValueError("not a GGUF file")
        version, n_tensors, n_kv = struct.unpack("<IQQ", self._take(20))
        if version not in (2, 3):
            raise ValueError(f"unsupported GGUF version {version}")
        self.version = version
        self.kv: dict[str, object] = {}
        self.kv_types: dict[str, int] = {}
        for _ in range(n_kv):
            key = self._rstr()
            (typ,) = struct.unpack("<I", self._take(4))
            self.kv[key] = self._rval(typ)
            self.kv_types[key] = typ
        infos = []
        for _ in range(n_tensors):
            name = self._rstr()
            (nd,) = struct.unpack("<I", self._take(4))
            ne = struct.unpack(f"<{nd}Q", self._take(8 * nd))
            typ, off = struct.unpack("<IQ", self._take(12))
            infos.append((name, ne, typ, off))
        align = int(self.kv.get("general.alignment", ALIGNMENT))
        base = _pad(self._p, align)
        self.tensors: dict[str, np.ndarray] = {}
        self.tensor_names: list[str] = []
        for name, ne, typ, off in infos:
            dt = _GGML2NP[typ]
            n = int(np.prod(ne))
            a = np.frombuffer(buf, dtype=dt, count=n, offset=base + off).reshape(tuple(ne)[::-1])
            self.tensors[name] = a
            self.tensor_names.append(name)

    def _take(self, n: int) -> bytes:
        b = self._b[self._p:self._p + n]
        if len(b) != n:
            raise ValueError("truncated GGUF file")
        self._p += n
        return b

    def _rstr(self) -> str:
        (n,) = struct.unpack("<Q", self._take(8))
        return self._take(n).decode("utf-8")

    def _rval(self, typ: int):
        if typ == T_STR:
            return self._rstr()
        if typ == T_ARR:
            et, n = struct.unpack("<IQ", self._take(12))
            return [self._rval(et) for _ in range(n)]
        fmt = _SCALAR_FMT[typ]
        return struct.unpack(fmt, self._take(struct.calcsize(fmt)))[0]
