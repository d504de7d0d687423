"""ctypes binding of ``libbigdl_host.so`` (C++ quantizers, re-layout, CPU quantized matmul).

Counterpart of the reference's ctypes layer over its closed CPU library
(python/llm/src/ipex_llm/ggml/model/llama/llama_cpp.py:946-1185) and of the helpers in
python/llm/src/ipex_llm/transformers/low_bit_linear.py:104-259.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

from .. import qtypes as Q
from ..qtypes import NF4, RTN_QTYPES, SYM_INT8_RTN, ASYM_INT4_RTN

_LIB = None
_LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libbigdl_host.so"


def lib():
    global _LIB
    if _LIB is None:
        if not _LIB_PATH.exists():
            from ..csrc.build import build_host
            build_host()
        L = ctypes.CDLL(str(_LIB_PATH))
        vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.bigdl_qk_size.argtypes = [ci]
        L.bigdl_type_size.argtypes = [ci]
        L.bigdl_quantize.argtypes = [vp, vp, ci, i64, i64, vp, ci]
        L.bigdl_dequantize.argtypes = [vp, vp, ci, i64, i64]
        L.bigdl_quantize_rtn.argtypes = [vp, vp, vp, ci, i64, i64, vp]
        L.bigdl_dequantize_rtn.argtypes = [vp, vp, vp, ci, i64, i64]
        L.bigdl_device_nbytes.argtypes = [ci, i64, i64]
        L.bigdl_device_nbytes.restype = i64
        L.bigdl_relayout_to_device.argtypes = [ci, vp, vp, i64, i64]
        L.bigdl_relayout_to_host.argtypes = [ci, vp, vp, i64, i64]
        L.bigdl_matmul_q.argtypes = [vp, vp, ci, i64, i64, i64, vp]
        L.bigdl_dequantize_rows.argtypes = [vp, i64, vp, ci, i64, i64, vp]
        L.bigdl_moe_indexes.argtypes = [vp, i64, i64, i64, vp, vp]
        _LIB = L
    return _LIB


def ggml_qk_size(qtype: int) -> int:
    return lib().bigdl_qk_size(qtype)


def ggml_type_size(qtype: int) -> int:
    return lib().bigdl_type_size(qtype)


def _cpu_contig(t: torch.Tensor, dtype=None) -> torch.Tensor:
    if t.device.type != "cpu":
        raise ValueError("host library needs CPU tensors")
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def quantized_nbytes(qtype: int, numel: int) -> int:
    return numel // ggml_qk_size(qtype) * ggml_type_size(qtype)


def quantize(weight: torch.Tensor, qtype: int, imatrix: torch.Tensor | None = None,
             scale_search: bool = False, shape_only: bool = False, device=None):
    """fp32 ``[..., K]`` tensor -> flat block buffer (uint8; nf4 is typed bf16 so that ZeRO-3
    sees one parameter dtype, as in reference low_bit_linear.py:132-138).
    RTN formats return ``(codes, scales_fp16)``."""
    k = weight.shape[-1]
    n = weight.numel()
    if qtype in RTN_QTYPES:
        rows = n // k
        code_dtype = torch.int8 if qtype == SYM_INT8_RTN else torch.uint8
        cols = k if qtype == SYM_INT8_RTN else k // 2
        codes = torch.empty((rows, cols), dtype=code_dtype, device=device)
        scale = torch.empty(rows * (2 if qtype == ASYM_INT4_RTN else 1), dtype=torch.float32, device=device)
        if not shape_only and device != "meta":
            w = _cpu_contig(weight, torch.float32)
            rc = lib().bigdl_quantize_rtn(w.data_ptr(), codes.data_ptr(), scale.data_ptr(), qtype, rows, k,
                                          imatrix.data_ptr() if imatrix is not None else None)
            if rc != 0:
                raise RuntimeError(f"bigdl_quantize_rtn failed ({rc})")
        return codes, scale.to(torch.float16)
    qk = ggml_qk_size(qtype)
    if qk <= 1:
        raise ValueError(f"qtype {qtype} is not a block-quantized format")
    if k % qk != 0:
        raise ValueError(f"Last dim of input tensor must be multiple of {qk}")
    nbytes = quantized_nbytes(qtype, n)
    if qtype == NF4:
        dst = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=device)
    else:
        dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if shape_only or device == "meta" or (device is not None and torch.device(device).type == "meta"):
        return dst
    w = _cpu_contig(weight, torch.float32)
    im = None
    if imatrix is not None:
        im = _cpu_contig(imatrix, torch.float32)
        if im.numel() != k:
            raise ValueError("imatrix must have in_features elements")
    rc = lib().bigdl_quantize(w.data_ptr(), dst.data_ptr(), qtype, n // k, k,
                              im.data_ptr() if im is not None else None, int(bool(scale_search)))
    if rc != 0:
        raise RuntimeError(f"bigdl_quantize failed ({rc})")
    return dst


def dequantize(buf: torch.Tensor, qtype: int, shape) -> torch.Tensor:
    """block buffer -> fp32 tensor of ``shape`` (host layout)."""
    buf = _cpu_contig(buf)
    shape = tuple(shape)
    k = shape[-1]
    n = 1
    for s in shape:
        n *= s
    out = torch.empty(shape, dtype=torch.float32)
    rc = lib().bigdl_dequantize(buf.data_ptr(), out.data_ptr(), qtype, n // k, k)
    if rc != 0:
        raise RuntimeError(f"bigdl_dequantize failed ({rc})")
    return out


def dequantize_rtn(codes: torch.Tensor, scale: torch.Tensor, qtype: int, shape) -> torch.Tensor:
    codes = _cpu_contig(codes)
    scale = _cpu_contig(scale, torch.float32)
    out = torch.empty(tuple(shape), dtype=torch.float32)
    rc = lib().bigdl_dequantize_rtn(codes.data_ptr(), scale.data_ptr(), out.data_ptr(), qtype, shape[0], shape[-1])
    if rc != 0:
        raise RuntimeError(f"bigdl_dequantize_rtn failed ({rc})")
    return out


def device_nbytes(qtype: int, n: int, k: int) -> int:
    return lib().bigdl_device_nbytes(qtype, n, k)


def relayout_to_device(buf: torch.Tensor, qtype: int, n: int, k: int) -> torch.Tensor:
    """host block order -> gfx950 tiled order (still a CPU tensor; the caller does the H2D copy)."""
    buf = _cpu_contig(buf)
    nb = device_nbytes(qtype, n, k)
    if nb < 0:
        raise ValueError(f"qtype {qtype} has no tiled device layout")
    out = torch.empty(nb, dtype=torch.uint8)
    rc = lib().bigdl_relayout_to_device(qtype, buf.data_ptr(), out.data_ptr(), n, k)
    if rc != 0:
        raise RuntimeError(f"relayout_to_device failed ({rc})")
    return out


def relayout_to_host(buf: torch.Tensor, qtype: int, n: int, k: int) -> torch.Tensor:
    buf = _cpu_contig(buf)
    out = torch.empty(quantized_nbytes(qtype, n * k), dtype=torch.uint8)
    rc = lib().bigdl_relayout_to_host(qtype, buf.data_ptr(), out.data_ptr(), n, k)
    if rc != 0:
        raise RuntimeError(f"relayout_to_host failed ({rc})")
    if qtype == NF4:
        out = out.view(torch.bfloat16)
    return out


def matmul_q(x: torch.Tensor, w: torch.Tensor, qtype: int, n: int, k: int) -> torch.Tensor:
    """``x[M, K] @ dequant(w)[N, K]^T`` on the CPU, fp32 in/out."""
    x2 = _cpu_contig(x.reshape(-1, k), torch.float32)
    w = _cpu_contig(w)
    y = torch.empty((x2.shape[0], n), dtype=torch.float32)
    rc = lib().bigdl_matmul_q(x2.data_ptr(), w.data_ptr(), qtype, x2.shape[0], n, k, y.data_ptr())
    if rc != 0:
        raise RuntimeError(f"bigdl_matmul_q failed ({rc})")
    return y.reshape(*x.shape[:-1], n)


def dequantize_rows(ids: torch.Tensor, w: torch.Tensor, qtype: int, dim: int, vocab: int) -> torch.Tensor:
    ids_c = _cpu_contig(ids, torch.int64)
    w = _cpu_contig(w)
    out = torch.empty((*ids.shape, dim), dtype=torch.float32)
    rc = lib().bigdl_dequantize_rows(ids_c.data_ptr(), ids_c.numel(), w.data_ptr(), qtype, dim, vocab, out.data_ptr())
    if rc != 0:
        raise IndexError("embedding index out of range" if rc == -2 else f"dequantize_rows failed ({rc})")
    return out


def moe_indexes(selected_experts: torch.Tensor, num_experts: int):
    """Host token->expert bucketing (reference op xe_linear.get_moe_indexes,
    python/llm/src/ipex_llm/transformers/models/qwen2_moe.py:404).  Returns ``(offsets[E+1], perm)``
    where ``perm`` lists flat ``token*top_k+slot`` indices grouped by expert."""
    sel = _cpu_contig(selected_experts, torch.int32)
    t, top_k = sel.shape
    offsets = torch.empty(num_experts + 1, dtype=torch.int64)
    perm = torch.empty(t * top_k, dtype=torch.int64)
    rc = lib().bigdl_moe_indexes(sel.data_ptr(), t, top_k, num_experts, offsets.data_ptr(), perm.data_ptr())
    if rc != 0:
        raise ValueError("expert id out of range")
    return offsets, perm


def t16_plane_nbytes(qtype: int, n: int, k: int):
    """Byte sizes of the planes of a tiled device buffer, in storage order (codes, high bits, scales, mins)."""
    np_, kp = (n + 15) // 16 * 16, (k + 127) // 128 * 128
    lo_bits = 8 if qtype in (Q.SYM_INT8, Q.FP8E4, Q.FP8E5) else 4
    hi_bits = {Q.SYM_INT5: 1, Q.ASYM_INT5: 1, Q.FP6: 2, Q.Q5_K: 1}.get(qtype, 0)
    has_min = qtype in (Q.ASYM_INT4, Q.ASYM_INT5, Q.WOQ_INT4, Q.Q4_K, Q.Q5_K)
    sizes = [np_ * kp * lo_bits // 8, np_ * kp * hi_bits // 8, np_ * (kp // 32) * 2, np_ * (kp // 32) * 2 if has_min else 0]
    assert sum(sizes) == device_nbytes(qtype, n, k)
    return sizes


def slice_t16_rows(buf: torch.Tensor, qtype: int, n: int, k: int, row0: int, nrows: int) -> torch.Tensor:
    """Rows [row0, row0 + nrows) of a tiled [n, k] buffer as a tiled [nrows, k] buffer (both multiples of 16).
    Planes are tile-row-major, so a row range is one contiguous byte range per plane (any device)."""
    if row0 % 16 or nrows % 16 or row0 + nrows > (n + 15) // 16 * 16:
        raise ValueError("row range must be 16-aligned and inside the matrix")
    np_ = (n + 15) // 16 * 16
    flat = buf.view(torch.uint8).reshape(-1)
    parts, off = [], 0
    for sz in t16_plane_nbytes(qtype, n, k):
        if sz:
            per_row = sz // np_
            parts.append(flat[off + row0 * per_row: off + (row0 + nrows) * per_row])
        off += sz
    return torch.cat(parts).contiguous()
