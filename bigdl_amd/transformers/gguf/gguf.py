"""GGUF v2 / v3 container: reader (memory-mapped), tensor dequantization, minimal writer.

Capability map to the reference python/llm/src/ipex_llm/transformers/gguf/gguf.py:31-424 (``GGUFReader`` header /
key-value / tensor-info parsing, ``GGUFTensorLoader`` with python dequantizers for q4_0 / q4_1 / q5_0 / q5_1 / q8_0 /
q6_k, ``GGUFFileLoader`` with ``config`` and ``tokenizer_pieces``).

The block formats of GGUF are the block formats of this package's host library (``sym_int4`` = q4_0, ``asym_int4`` =
q4_1, ``sym_int5`` = q5_0, ``asym_int5`` = q5_1, ``sym_int8`` = q8_0, q2_k / q4_k / q5_k / q6_k), so dequantization
is one call into the C++ dequantizer on the mapped bytes -- no python unpacking loops -- and more tensor types are
readable than in the reference (all k-quants except q3_k, bf16).
"""
from __future__ import annotations

import mmap
import struct
from typing import Dict, List

import numpy as np
import torch

from ... import qtypes as Q
from ...ops import host
from ...utils.common import invalidInputError

GGUF_MAGIC = 0x46554747
# metadata value types
_SCALARS = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}
T_STRING, T_ARRAY = 8, 9

# ggml tensor types
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q4_1, GGML_Q5_0, GGML_Q5_1, GGML_Q8_0 = 0, 1, 2, 3, 6, 7, 8
GGML_Q2_K, GGML_Q3_K, GGML_Q4_K, GGML_Q5_K, GGML_Q6_K, GGML_BF16 = 10, 11, 12, 13, 14, 30
GGML_TO_QTYPE = {GGML_Q4_0: Q.SYM_INT4, GGML_Q4_1: Q.ASYM_INT4, GGML_Q5_0: Q.SYM_INT5, GGML_Q5_1: Q.ASYM_INT5,
                 GGML_Q8_0: Q.SYM_INT8, GGML_Q2_K: Q.Q2_K, GGML_Q4_K: Q.Q4_K, GGML_Q5_K: Q.Q5_K, GGML_Q6_K: Q.Q6_K}
QTYPE_TO_GGML = {v: k for k, v in GGML_TO_QTYPE.items()}
GGML_NAMES = {0: "f32", 1: "f16", 2: "q4_0", 3: "q4_1", 6: "q5_0", 7: "q5_1", 8: "q8_0", 10: "q2_k", 11: "q3_k",
              12: "q4_k", 13: "q5_k", 14: "q6_k", 30: "bf16"}


def _tensor_nbytes(ggml_type: int, n_elems: int) -> int:
    if ggml_type == GGML_F32:
        return n_elems * 4
    if ggml_type in (GGML_F16, GGML_BF16):
        return n_elems * 2
    qt = GGML_TO_QTYPE.get(ggml_type)
    invalidInputError(qt is not None, f"unsupported ggml tensor type {GGML_NAMES.get(ggml_type, ggml_type)}")
    return n_elems // host.ggml_qk_size(qt) * host.ggml_type_size(qt)


class GGUFReader:
    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._pos = 0
        magic, self.version = self._unpack("<II")
        invalidInputError(magic == GGUF_MAGIC, "not a GGUF file")
        invalidInputError(self.version in (2, 3), f"unsupported GGUF version {self.version}")
        n_tensors, n_kv = self._unpack("<QQ")
        self.metadata: Dict[str, object] = {}
        for _ in range(n_kv):
            key = self._string()
            (vt,) = self._unpack("<I")
            self.metadata[key] = self._value(vt)
        self.tensor_infos: Dict[str, dict] = {}
        for _ in range(n_tensors):
            name = self._string()
            (nd,) = self._unpack("<I")
            dims = list(self._unpack("<" + "Q" * nd)) if nd else []
            ggml_type, offset = self._unpack("<IQ")
            self.tensor_infos[name] = dict(dims=dims, shape=tuple(reversed(dims)), type=ggml_type, offset=offset)
        align = int(self.metadata.get("general.alignment", 32))
        self.data_offset = (self._pos + align - 1) // align * align

    def _unpack(self, fmt):
        size = struct.calcsize(fmt)
        out = struct.unpack_from(fmt, self._mm, self._pos)
        self._pos += size
        return out

    def _string(self):
        (n,) = self._unpack("<Q")
        s = bytes(self._mm[self._pos:self._pos + n])
        self._pos += n
        return s.decode("utf-8", errors="replace")

    def _value(self, vt):
        if vt in _SCALARS:
            return self._unpack(_SCALARS[vt])[0]
        if vt == T_STRING:
            return self._string()
        if vt == T_ARRAY:
            et, n = self._unpack("<IQ")
            if et in _SCALARS:
                fmt = _SCALARS[et]
                size = struct.calcsize(fmt)
                arr = np.frombuffer(self._mm, dtype=np.dtype(fmt[1:]).newbyteorder("<"), count=n, offset=self._pos)
                self._pos += size * n
                return arr.tolist()
            return [self._value(et) for _ in range(n)]
        invalidInputError(False, f"unknown GGUF metadata type {vt}")

    def raw(self, name: str) -> torch.Tensor:
        """The tensor's bytes as a uint8 tensor backed by the file mapping (no copy)."""
        info = self.tensor_infos[name]
        n = int(np.prod(info["dims"])) if info["dims"] else 1
        nb = _tensor_nbytes(info["type"], n)
        start = self.data_offset + info["offset"]
        arr = np.frombuffer(self._mm, dtype=np.uint8, count=nb, offset=start)
        return torch.from_numpy(arr)

    def close(self):
        try:
            self._mm.close()
        except BufferError:
            pass          # tensors still view the mapping; it is released with them
        self._f.close()


class GGUFTensorLoader:
    def __init__(self, reader: GGUFReader):
        self.reader = reader

    def names(self) -> List[str]:
        return list(self.reader.tensor_infos)

    def info(self, name):
        return self.reader.tensor_infos[name]

    def quantized(self, name):
        """(raw block bytes, qtype, shape) for block-quantized tensors, else None."""
        info = self.reader.tensor_infos[name]
        qt = GGML_TO_QTYPE.get(info["type"])
        if qt is None:
            return None
        return self.reader.raw(name), qt, info["shape"]

    def load(self, name, dtype=torch.float32) -> torch.Tensor:
        info = self.reader.tensor_infos[name]
        raw, shape, t = self.reader.raw(name), info["shape"], info["type"]
        if t == GGML_F32:
            out = raw.view(torch.float32).view(shape)
        elif t == GGML_F16:
            out = raw.view(torch.float16).view(shape)
        elif t == GGML_BF16:
            out = raw.view(torch.bfloat16).view(shape)
        else:
            qt = GGML_TO_QTYPE.get(t)
            invalidInputError(qt is not None, f"tensor {name}: unsupported ggml type {GGML_NAMES.get(t, t)}")
            rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
            out = host.dequantize(raw, qt, (rows, shape[-1])).view(shape)
        return out.to(dtype)

    def load_while_process(self, process):
        for name in self.names():
            process(name, self.load(name))


class GGUFFileLoader:
    def __init__(self, fpath: str):
        self.reader = GGUFReader(fpath)
        self.config = self.reader.metadata
        self.tensor_loader = GGUFTensorLoader(self.reader)

    def tokenizer_dict(self):
        """The ``tokenizer.*`` metadata in the shape HuggingFace's gguf tokenizer converters expect."""
        c = self.config
        out = {"tokenizer_type": c.get("tokenizer.ggml.model", "llama"), "tokens": c.get("tokenizer.ggml.tokens", []),
               "scores": c.get("tokenizer.ggml.scores"), "token_type": c.get("tokenizer.ggml.token_type"),
               "merges": c.get("tokenizer.ggml.merges"), "bos_token_id": c.get("tokenizer.ggml.bos_token_id"),
               "eos_token_id": c.get("tokenizer.ggml.eos_token_id"), "unk_token_id": c.get("tokenizer.ggml.unknown_token_id"),
               "pad_token_id": c.get("tokenizer.ggml.padding_token_id"),
               "add_prefix_space": c.get("tokenizer.ggml.add_space_prefix", True)}
        return {k: v for k, v in out.items() if v is not None}

    def tokenizer_pieces(self):
        """(piece, score, type) triples of a sentencepiece-style vocabulary."""
        c = self.config
        toks = c["tokenizer.ggml.tokens"]
        scores = c.get("tokenizer.ggml.scores") or [0.0] * len(toks)
        types = c.get("tokenizer.ggml.token_type") or [1] * len(toks)
        return list(zip(toks, scores, types))


class GGUFWriter:
    """Minimal GGUF v3 writer (metadata + tensors) used to export models and to build test fixtures."""

    def __init__(self, alignment: int = 32):
        self.alignment = alignment
        self.kv = []
        self.tensors = []

    def add(self, key, value, vtype=None):
        self.kv.append((key, value, vtype))

    def add_tensor(self, name, tensor: torch.Tensor, ggml_type: int = GGML_F32):
        shape = tuple(tensor.shape)
        if ggml_type == GGML_F32:
            data = tensor.detach().float().contiguous().numpy().tobytes()
        elif ggml_type == GGML_F16:
            data = tensor.detach().half().contiguous().numpy().tobytes()
        else:
            qt = GGML_TO_QTYPE[ggml_type]
            q = host.quantize(tensor.detach().float().reshape(-1, shape[-1]).contiguous(), qt)
            data = q.view(torch.uint8).numpy().tobytes()
        self.tensors.append((name, shape, ggml_type, data))

    @staticmethod
    def _s(s: str) -> bytes:
        b = s.encode("utf-8")
        return struct.pack("<Q", len(b)) + b

    def _val(self, v, vtype=None) -> bytes:
        if isinstance(v, bool):
            return struct.pack("<I?", 7, v)
        if isinstance(v, int):
            vt = vtype or 4
            return struct.pack("<I", vt) + struct.pack(_SCALARS[vt], v)
        if isinstance(v, float):
            return struct.pack("<If", 6, v)
        if isinstance(v, str):
            return struct.pack("<I", T_STRING) + self._s(v)
        if isinstance(v, (list, tuple)):
            if len(v) and isinstance(v[0], str):
                body = b"".join(self._s(x) for x in v)
                et = T_STRING
            elif len(v) and isinstance(v[0], float):
                body = struct.pack(f"<{len(v)}f", *v)
                et = 6
            else:
                body = struct.pack(f"<{len(v)}i", *v)
                et = 5
            return struct.pack("<IIQ", T_ARRAY, et, len(v)) + body
        raise TypeError(f"unsupported metadata value {type(v)}")

    def write(self, path: str):
        head = struct.pack("<IIQQ", GGUF_MAGIC, 3, len(self.tensors), len(self.kv) + 1)
        head += self._s("general.alignment") + struct.pack("<II", 4, self.alignment)
        for k, v, vt in self.kv:
            head += self._s(k) + self._val(v, vt)
        off = 0
        infos = b""
        blobs = []
        for name, shape, t, data in self.tensors:
            dims = tuple(reversed(shape))
            infos += self._s(name) + struct.pack("<I", len(dims)) + struct.pack("<" + "Q" * len(dims), *dims)
            infos += struct.pack("<IQ", t, off)
            pad = (-len(data)) % self.alignment
            blobs.append(data + b"\0" * pad)
            off += len(data) + pad
        head += infos
        pad = (-len(head)) % self.alignment
        with open(path, "wb") as f:
            f.write(head + b"\0" * pad)
            for b in blobs:
                f.write(b)
