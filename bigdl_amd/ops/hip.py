"""ctypes binding of ``libbigdl_hip.so`` — the gfx950 kernels.

Every wrapper validates shapes/dtypes/devices on the host before the launch (a kernel that
faults can take the whole node down), launches on PyTorch's current HIP stream (so calls are
capturable into a hipGraph via ``torch.cuda.graph``) and raises if the native launcher rejects
the call.  There is no silent fallback: on a GPU box a missing library is an error.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

from .. import qtypes as Q

_LIB = None
# BIGDL_HIP_LIB: another build of the library (A/B measurements of a kernel change inside one process tree)
_LIB_PATH = Path(os.environ.get("BIGDL_HIP_LIB") or Path(__file__).resolve().parent.parent / "lib" / "libbigdl_hip.so")

OUT_F16, OUT_BF16, OUT_F32 = 0, 1, 2
ACT = {None: 0, "none": 0, "silu": 1, "swish": 1, "gelu": 2, "gelu_tanh": 3, "gelu_pytorch_tanh": 3,
       "gelu_new": 3, "gelu_fast": 3, "relu": 4}
_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}


def available() -> bool:
    return torch.cuda.is_available() and _LIB_PATH.exists()


def lib():
    global _LIB
    if _LIB is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -m bigdl_amd.csrc.build` "
                "(hipcc cross-compiles gfx950 without a GPU)")
        L = ctypes.CDLL(str(_LIB_PATH))
        vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.bigdl_gemv_lowbit.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, vp]
        L.bigdl_gemv_f16.argtypes = [vp] * 8 + [cf, cf] + [ci] * 7 + [vp]
        L.bigdl_gemv_cu_rows.argtypes = [ci, ci]
        L.bigdl_gemv_cu_rows.restype = None
        cu_rows = os.environ.get("BIGDL_GEMV_CU_ROWS")
        if cu_rows:
            lo_, hi_ = (int(v) for v in cu_rows.split(","))
            L.bigdl_gemv_cu_rows(lo_, hi_)
        L.bigdl_gemv_f16_tune.argtypes = [ci]
        L.bigdl_gemv_f16_tune.restype = None
        L.bigdl_gemv_lowbit_ln.argtypes = [vp] * 8 + [cf] + [ci] * 8 + [vp] * 5
        L.bigdl_attn_decode.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, cf, cf] + [ci] * 14 + [vp, vp, vp]
        L.bigdl_kv_int4_append.argtypes = [vp, vp, vp, vp, vp] + [ci] * 7 + [i64, vp]
        L.bigdl_attn_decode_ws_floats.argtypes = [ci] * 6
        L.bigdl_attn_decode_ws_floats.restype = i64
        L.bigdl_attn_decode_counters.argtypes = [ci] * 3
        L.bigdl_rope_append.argtypes = [vp, vp, vp, vp, vp, vp, cf] + [ci] * 10 + [vp]
        L.bigdl_rms_norm.argtypes = [vp, vp, vp, vp, vp, i64, ci, i64, cf, cf, ci, vp]
        L.bigdl_quant_rows_e4m3.argtypes = [vp, vp, vp, i64, ci, i64, i64, vp]
        L.bigdl_bf16_to_f16.argtypes = [vp, vp, i64, vp]
        L.bigdl_layer_norm.argtypes = [vp, vp, vp, vp, i64, ci, i64, cf, ci, vp]
        L.bigdl_act_mul.argtypes = [vp, vp, vp, i64, ci, i64, i64, i64, ci, ci, vp]
        L.bigdl_silu_mul_bwd.argtypes = [vp, vp, vp, vp, vp, i64, ci, i64, i64, i64, ci, vp]
        L.bigdl_rms_norm_bwd.argtypes = [vp, vp, vp, vp, i64, ci, i64, i64, cf, cf, ci, vp]
        L.bigdl_softmax_inplace.argtypes = [vp, i64, ci, cf, ci, vp]
        L.bigdl_rope_inplace.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, i64, i64, ci, ci, ci, ci, ci, ci, ci, cf, ci, vp]
        L.bigdl_to_e5m2.argtypes = [vp, vp, i64, ci, i64, i64, ci, vp]
        L.bigdl_from_e5m2.argtypes = [vp, vp, i64, ci, i64, i64, ci, vp]
        L.bigdl_repetition_penalty.argtypes = [vp, vp, ci, ci, ci, cf, ci, vp]
        L.bigdl_argmax.argtypes = [vp, vp, i64, ci, i64, ci, ci, vp, vp, vp, ci, ci, vp]
        L.bigdl_dequant_t16.argtypes = [vp, vp, ci, ci, i64, ci, ci, ci, ci, vp]
        L.bigdl_dequant_rows.argtypes = [vp, vp, vp, i64, i64, ci, ci, i64, ci, vp]
        L.bigdl_gemv_lowbit_moe.argtypes = [vp, vp, vp, vp, vp, vp, vp, cf, cf, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
        L.bigdl_gemv_lowbit_moe_sum.argtypes = [vp] * 5 + [ci] * 8 + [vp]
        L.bigdl_gemm_lowbit.argtypes = [vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, ci, ci, ci, ci, vp]
        L.bigdl_gemm_lowbit_splitk.argtypes = [ci, ci, ci]
        L.bigdl_gemm_prompt.argtypes = [vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, ci, ci, ci, ci, vp]
        L.bigdl_gemm_prompt_pair.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
        L.bigdl_gemm_prompt_slabs.argtypes = [ci, ci, ci, ci, ci, ci]
        L.bigdl_gemm_prompt_supported.argtypes = [ci, ci]
        L.bigdl_gemm_prompt_plan.argtypes = [ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.bigdl_gemm_prompt_force.argtypes = [ci, ci]
        L.bigdl_gemm_prompt_force.restype = None
        L.bigdl_gemm_lowbit_ex.argtypes = [vp] * 8 + [cf, cf, vp, vp, vp, i64] + [ci] * 8 + [vp]
        L.bigdl_gemm_lowbit_ex_splitk.argtypes = [ci, ci, ci, ci]
        L.bigdl_gemm_lowbit_bm.argtypes = [ci]
        L.bigdl_gemm_lowbit_grouped.argtypes = [vp] * 8 + [ci] * 12 + [vp]
        L.bigdl_moe_block_table.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
        L.bigdl_lora_down.argtypes = [vp, vp, vp, ci, ci, ci, i64, i64, ci, cf, ci, vp]
        L.bigdl_lora_up.argtypes = [vp, vp, vp, vp, ci, ci, ci, i64, i64, i64, ci, cf, ci, ci, vp]
        L.bigdl_moe_route.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, i64, ci, ci, vp]
        L.bigdl_moe_route_grouped.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, i64, ci, ci, vp, ci, ci, ci, vp]
        L.bigdl_moe_combine.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, i64, vp]
        L.bigdl_moe_norm_route.argtypes = [vp] * 6 + [ci, ci, i64, cf, cf] + [ci] * 6 + [vp, ci, ci, ci, vp]
        L.bigdl_rwkv_time_shift.argtypes = [vp, vp, vp, vp, i64, ci, ci, ci, vp]
        L.bigdl_rwkv4_wkv.argtypes = [vp] * 8 + [ci] * 5 + [vp]
        L.bigdl_rwkv5_wkv.argtypes = [vp] * 7 + [ci] * 4 + [vp]
        L.bigdl_prefill_rope_append.argtypes = [vp, vp, vp, vp, vp, vp] + [ci] * 10 + [i64, i64, ci, vp]
        L.bigdl_attn_prefill.argtypes = [vp, vp, vp, vp, vp] + [ci] * 6 + [i64] * 14 + [vp, cf, cf, ci, ci, ci, ci, vp]
        _LIB = L
    return _LIB


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: native launcher rejected the call (code {rc})")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise ValueError("expected tensors on the HIP device")


def _out_type(dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {dtype}")


def gemv_lowbit(x, w, qtype, out_features, in_features, *, w2=None, bias=None, bias2=None, residual=None,
                gamma=None, eps=1e-6, gamma_offset=0.0, act=None, out=None, out_dtype=None, att_partials=None):
    """``y = act(rmsnorm?(x) @ dequant(w)^T + bias) [* (x @ dequant(w2)^T + bias2)] [+ residual]``

    x: [M, K] fp16 (M <= 64, or <= 32 with w2), row stride multiple of 8.  w/w2: T16 device buffers.
    """
    _need_cuda(x, w, w2, bias, bias2, residual, gamma, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D fp16 tensor with contiguous rows")
    m, k = x.shape
    if k != in_features:
        raise ValueError(f"x has {k} features, weight expects {in_features}")
    if qtype not in Q.T16_QTYPES:
        raise ValueError(f"qtype {qtype} has no fused GEMV kernel")
    from .host import device_nbytes
    need = device_nbytes(qtype, out_features, in_features)
    for ww in (w, w2):
        if ww is not None and ww.numel() * ww.element_size() != need:
            raise ValueError(f"weight buffer has {ww.numel() * ww.element_size()} bytes, expected {need} "
                             f"for [{out_features}, {in_features}] qtype {qtype}")
    out_dtype = out_dtype or (out.dtype if out is not None else torch.float16)
    if out is None:
        out = torch.empty((m, out_features), dtype=out_dtype, device=x.device)
    if out.shape != (m, out_features) or out.stride(1) != 1 or out.dtype != out_dtype:
        raise ValueError("bad out tensor")
    for b in (bias, bias2):
        if b is not None and (b.dtype != torch.float16 or b.numel() != out_features or not b.is_contiguous()):
            raise ValueError("bias must be contiguous fp16 [out_features]")
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype
                                 or residual.stride() != out.stride()):
        raise ValueError("residual must match out")
    if gamma is not None and (gamma.dtype != torch.float16 or gamma.numel() != k or not gamma.is_contiguous()):
        raise ValueError("gamma must be contiguous fp16 [in_features]")
    att = (None, 0, 0, 0, 0, 0, 0)
    if att_partials is not None:
        # (ws fp32, splits, head_dim, group, n_kv_heads, q_len, rows_alloc): x is combined from the split-KV
        # attention partials while it is staged (x itself is then only a shape/stride carrier)
        ws_t, splits, hd, grp, nkv, qlen, rows_alloc = att_partials
        _need_cuda(ws_t)
        if ws_t.dtype != torch.float32 or ws_t.numel() < (m // qlen) * nkv * splits * rows_alloc * (hd + 4):
            raise ValueError("attention partial workspace too small")
        att = (_ptr(ws_t), splits, hd, grp, nkv, qlen, rows_alloc)
    rc = lib().bigdl_gemv_lowbit(_ptr(x), _ptr(w), _ptr(w2), _ptr(out), _ptr(bias), _ptr(bias2), _ptr(residual),
                                 _ptr(gamma), eps, gamma_offset, m, out_features, k, x.stride(0), out.stride(0),
                                 qtype, ACT[act], _out_type(out_dtype), *att, _stream())
    _check(rc, "gemv_lowbit")
    return out


def gemv_f16_ok(m, n, k):
    """shapes the fp16 skinny GEMM takes (csrc/hip/gemv_f16.hip)"""
    return 0 < m <= 16 and n % 16 == 0 and k % 64 == 0


def gemv_f16_tune(rows_form=1):
    """measurement hook: ``rows_form=0`` sends 1 - 4 rows to the MFMA form too (benchmark/gemv_f16_probe.py)"""
    lib().bigdl_gemv_f16_tune(int(rows_form))


def gemv_f16(x, w, *, w2=None, bias=None, bias2=None, residual=None, gamma=None, eps=1e-6, gamma_offset=0.0, act=None,
             out=None, out_dtype=None):
    """``y = act(rmsnorm?(x) @ w^T + bias) [* (x @ w2^T + bias2)] [+ residual]`` for plain fp16 weights [N, K]
    (row-major, contiguous), M <= 16 rows of x."""
    _need_cuda(x, w, w2, bias, bias2, residual, gamma, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D fp16 tensor with contiguous rows")
    m, k = x.shape
    for ww in (w, w2):
        if ww is not None and (ww.dtype != torch.float16 or ww.dim() != 2 or not ww.is_contiguous() or ww.shape != w.shape
                               or ww.shape[1] != k):
            raise ValueError("weights must be contiguous fp16 [out_features, in_features] matching x")
    n = w.shape[0]
    if not gemv_f16_ok(m, n, k):
        raise ValueError(f"gemv_f16 takes M <= 16, N % 16 == 0, K % 64 == 0; got M={m} N={n} K={k}")
    out_dtype = out_dtype or (out.dtype if out is not None else torch.float16)
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=x.device)
    if out.shape != (m, n) or out.stride(1) != 1 or out.dtype != out_dtype:
        raise ValueError("bad out tensor")
    for b in (bias, bias2):
        if b is not None and (b.dtype != torch.float16 or b.numel() != n or not b.is_contiguous()):
            raise ValueError("bias must be contiguous fp16 [out_features]")
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype
                                 or residual.stride() != out.stride()):
        raise ValueError("residual must match out")
    if gamma is not None and (gamma.dtype != torch.float16 or gamma.numel() != k or not gamma.is_contiguous()):
        raise ValueError("gamma must be contiguous fp16 [in_features]")
    rc = lib().bigdl_gemv_f16(_ptr(x), _ptr(w), _ptr(w2), _ptr(out), _ptr(bias), _ptr(bias2), _ptr(residual), _ptr(gamma),
                              eps, gamma_offset, m, n, k, x.stride(0), out.stride(0), ACT[act], _out_type(out_dtype),
                              _stream())
    _check(rc, "gemv_f16")
    return out


def gemv_lowbit_moe(x, w, qtype, n_experts, out_features, in_features, slot_expert, *, w2=None, slot_scale=None,
                    slot_xdiv=1, gamma=None, eps=1e-6, gamma_offset=0.0, act=None, out=None):
    """Routed-expert products: ``out[s] = slot_scale[s] * f(x[s // slot_xdiv] @ dequant(W[slot_expert[s]])^T)``.

    w / w2: T16 buffer of the experts' matrices stacked by rows, [n_experts * out_features, in_features].  slot_expert: int32 [n_slots] on
    the device (values < n_experts; negative = skipped, zero row)."""
    _need_cuda(x, w, w2, slot_expert, slot_scale, gamma, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1 or x.shape[1] != in_features:
        raise ValueError("x must be a 2-D fp16 tensor [rows, in_features] with contiguous rows")
    if qtype not in Q.T16_QTYPES:
        raise ValueError(f"qtype {qtype} has no fused GEMV kernel")
    from .host import device_nbytes
    if out_features % 16:
        raise ValueError("expert out_features must be a multiple of 16")
    need = device_nbytes(qtype, out_features * n_experts, in_features)
    for ww in (w, w2):
        if ww is not None and ww.numel() * ww.element_size() != need:
            raise ValueError("expert weight buffer size does not match n_experts x [out_features, in_features]")
    if slot_expert.dtype != torch.int32 or not slot_expert.is_contiguous():
        raise ValueError("slot_expert must be contiguous int32")
    n_slots = slot_expert.numel()
    if (n_slots + slot_xdiv - 1) // slot_xdiv > x.shape[0]:
        raise ValueError("x has too few rows for the slots")
    if slot_scale is not None and (slot_scale.dtype != torch.float32 or slot_scale.numel() != n_slots
                                   or not slot_scale.is_contiguous()):
        raise ValueError("slot_scale must be contiguous fp32 [n_slots]")
    if gamma is not None and (gamma.dtype != torch.float16 or gamma.numel() != in_features or not gamma.is_contiguous()):
        raise ValueError("gamma must be contiguous fp16 [in_features]")
    if out is None:
        out = torch.empty((n_slots, out_features), dtype=torch.float16, device=x.device)
    if out.shape != (n_slots, out_features) or out.stride(1) != 1:
        raise ValueError("bad out tensor")
    rc = lib().bigdl_gemv_lowbit_moe(_ptr(x), _ptr(w), _ptr(w2), _ptr(out), _ptr(slot_expert), _ptr(slot_scale),
                                     _ptr(gamma), eps, gamma_offset, n_slots, slot_xdiv, n_experts, out_features,
                                     in_features, x.stride(0), out.stride(0), qtype, ACT[act], _out_type(out.dtype),
                                     _stream())
    _check(rc, "gemv_lowbit_moe")
    return out


def gemv_lowbit_moe_sum_ok(top_k, in_features) -> bool:
    """whether :func:`gemv_lowbit_moe_sum` takes a product of this shape (x of one token has to fit the LDS staging)"""
    return top_k <= 8 and in_features % 128 == 0 and (top_k * in_features + 136) * 2 <= 100 * 1024


def gemv_lowbit_moe_sum(x, w, qtype, n_experts, out_features, in_features, slot_expert, top_k, *, residual=None, out=None):
    """Routed products summed per token inside the launch: ``out[t] = residual[t] + sum_j x[t * top_k + j] @
    dequant(W[slot_expert[t * top_k + j]])^T``.  x [tokens * top_k, in_features] fp16 contiguous, already scaled by the
    routing weights (zero rows for negative expert ids); ``out`` may be ``residual`` itself."""
    _need_cuda(x, w, slot_expert, residual, out)
    if x.dim() != 2 or x.dtype != torch.float16 or not x.is_contiguous() or x.shape[1] != in_features:
        raise ValueError("x must be a contiguous fp16 tensor [tokens * top_k, in_features]")
    if qtype not in Q.T16_QTYPES:
        raise ValueError(f"qtype {qtype} has no fused GEMV kernel")
    if out_features % 16 or not gemv_lowbit_moe_sum_ok(top_k, in_features):
        raise ValueError("gemv_lowbit_moe_sum: out_features % 16, in_features % 128, top_k <= 8, top_k * in_features <= 51064")
    from .host import device_nbytes
    if w.numel() * w.element_size() != device_nbytes(qtype, out_features * n_experts, in_features):
        raise ValueError("expert weight buffer size does not match n_experts x [out_features, in_features]")
    if slot_expert.dtype != torch.int32 or not slot_expert.is_contiguous() or slot_expert.numel() != x.shape[0] \
            or x.shape[0] % top_k:
        raise ValueError("slot_expert must be contiguous int32 [tokens * top_k]")
    t = x.shape[0] // top_k
    if out is None:
        out = torch.empty((t, out_features), dtype=torch.float16 if residual is None else residual.dtype, device=x.device)
    if out.shape != (t, out_features) or out.stride(1) != 1:
        raise ValueError("bad out tensor")
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype or residual.stride() != out.stride()):
        raise ValueError("residual must match out")
    rc = lib().bigdl_gemv_lowbit_moe_sum(_ptr(x), _ptr(w), _ptr(out), _ptr(residual), _ptr(slot_expert), t, top_k, n_experts,
                                         out_features, in_features, out.stride(0), qtype, _out_type(out.dtype), _stream())
    _check(rc, "gemv_lowbit_moe_sum")
    return out


_GEMM_WS = {}


def _gemm_workspace(device, floats):
    """fp32 scratch for split-K partials, grown on demand and kept per device (stream-ordered reuse: every GEMM
    call consumes its partials before returning control to the stream's next kernel)."""
    ws = _GEMM_WS.get(device)
    if ws is None or ws.numel() < floats:
        ws = _GEMM_WS[device] = torch.empty(max(floats, 1 << 22), dtype=torch.float32, device=device)
    return ws


def gemm_lowbit(x, w, qtype, out_features, in_features, *, bias=None, residual=None, act=None, out=None,
                out_dtype=None):
    """``y = act(x @ dequant(w)^T + bias) [+ residual]`` on the MFMA GEMM kernel: x [M, K] fp16 with any M (meant
    for 16 < M <= 512; larger M is better served by dequantize + hipBLASLt).  w: T16 device buffer."""
    _need_cuda(x, w, bias, residual, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D fp16 tensor with contiguous rows")
    m, k = x.shape
    if k != in_features:
        raise ValueError(f"x has {k} features, weight expects {in_features}")
    if qtype not in Q.T16_QTYPES:
        raise ValueError(f"qtype {qtype} has no fused GEMM kernel")
    from .host import device_nbytes
    if w.numel() * w.element_size() != device_nbytes(qtype, out_features, in_features):
        raise ValueError("weight buffer size does not match the tiled layout")
    out_dtype = out_dtype or (out.dtype if out is not None else torch.float16)
    if out is None:
        out = torch.empty((m, out_features), dtype=out_dtype, device=x.device)
    if out.shape != (m, out_features) or out.stride(1) != 1 or out.dtype != out_dtype:
        raise ValueError("bad out tensor")
    if bias is not None and (bias.dtype != torch.float16 or bias.numel() != out_features or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous fp16 [out_features]")
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype
                                 or residual.stride() != out.stride()):
        raise ValueError("residual must match out")
    L = lib()
    sk = L.bigdl_gemm_lowbit_splitk(m, out_features, k)
    ws, ws_n = None, 0
    if sk > 1:
        ws_n = sk * m * out_features
        ws = _gemm_workspace(x.device, ws_n)
        ws_n = ws.numel()
    rc = L.bigdl_gemm_lowbit(_ptr(x), _ptr(w), _ptr(out), _ptr(bias), _ptr(residual), _ptr(ws), ws_n, m, out_features, k,
                             x.stride(0), out.stride(0), qtype, ACT[act], _out_type(out_dtype), _stream())
    _check(rc, "gemm_lowbit")
    return out


def gemm_prompt_supported(qtype, in_features) -> bool:
    """every tiled format, K a multiple of 128 and at least 256"""
    return bool(lib().bigdl_gemm_prompt_supported(int(qtype), int(in_features)))


def gemm_prompt_plan(m, out_features, in_features):
    """(tile configuration, K splits) the prompt GEMM will use: 0 = 256 x 256, 1 = 256 x 192, 2 = 128 x 128 tiles"""
    cfg, sk = ctypes.c_int(0), ctypes.c_int(0)
    lib().bigdl_gemm_prompt_plan(int(m), int(out_features), int(in_features), ctypes.byref(cfg), ctypes.byref(sk))
    return cfg.value, sk.value


def gemm_prompt_force(cfg=-1, splitk=0):
    """pin the tile configuration / K splits of the following ``gemm_prompt`` calls (tests, sweeps); no arguments: planned"""
    lib().bigdl_gemm_prompt_force(int(cfg), int(splitk))


def gemm_prompt(x, w, qtype, out_features, in_features, *, bias=None, residual=None, act=None, out=None,
                out_dtype=None):
    """``y = act(x @ dequant(w)^T + bias) [+ residual]`` on the prompt-sized fused MFMA GEMM (csrc/hip/
    gemm_prompt_kernel.h: 256 x 256 / 256 x 192 / 128 x 128 tiles, x through ``global_load_lds``, weights expanded once
    per workgroup into LDS): x [M, K] fp16, meant for M > 256; K % 128 == 0.  w: T16 device buffer.  The reference's
    one native op for every row count: python/llm/src/ipex_llm/transformers/low_bit_linear.py:657-668."""
    _need_cuda(x, w, bias, residual, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D fp16 tensor with contiguous rows")
    m, k = x.shape
    if k != in_features:
        raise ValueError(f"x has {k} features, weight expects {in_features}")
    if not gemm_prompt_supported(qtype, k):
        raise ValueError(f"qtype {qtype} / K = {k} has no prompt-sized fused GEMM kernel")
    from .host import device_nbytes
    if w.numel() * w.element_size() != device_nbytes(qtype, out_features, in_features):
        raise ValueError("weight buffer size does not match the tiled layout")
    out_dtype = out_dtype or (out.dtype if out is not None else torch.float16)
    if out is None:
        out = torch.empty((m, out_features), dtype=out_dtype, device=x.device)
    if out.shape != (m, out_features) or out.stride(1) != 1 or out.dtype != out_dtype:
        raise ValueError("bad out tensor")
    if bias is not None and (bias.dtype != torch.float16 or bias.numel() != out_features or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous fp16 [out_features]")
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype
                                 or residual.stride() != out.stride()):
        raise ValueError("residual must match out")
    L = lib()
    aligned = (out.data_ptr() % 16 == 0) and (residual is None or residual.data_ptr() % 16 == 0)
    slabs = L.bigdl_gemm_prompt_slabs(m, out_features, k, out.stride(0), _out_type(out_dtype), int(aligned))
    ws, ws_n = None, 0
    if slabs > 0:
        ws = _gemm_workspace(x.device, slabs * m * out_features)
        ws_n = ws.numel()
    rc = L.bigdl_gemm_prompt(_ptr(x), _ptr(w), _ptr(out), _ptr(bias), _ptr(residual), _ptr(ws), ws_n, m, out_features, k,
                             x.stride(0), out.stride(0), qtype, ACT[act], _out_type(out_dtype), _stream())
    _check(rc, "gemm_prompt")
    return out


def gemm_prompt_pair(x, w, w2, qtype, out_features, in_features, *, bias=None, bias2=None, act="silu", out=None,
                     out_dtype=None):
    """``act(x @ dequant(w)^T + bias) * (x @ dequant(w2)^T + bias2)``: the gated MLP's two input projections and their
    activation product in ONE launch of the prompt-sized fused GEMM (every workgroup computes the same columns of both
    products and combines them in its epilogue: neither product is written to memory).  Role of the reference's
    ``xe_linear.mlp_forward_xpu`` (python/llm/src/ipex_llm/transformers/models/common.py:118-133), which is limited
    to one row.  x [M, K] fp16; w, w2: T16 buffers of one format, [out_features, in_features], out_features % 16 == 0."""
    _need_cuda(x, w, w2, bias, bias2, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D fp16 tensor with contiguous rows")
    m, k = x.shape
    if k != in_features or out_features % 16:
        raise ValueError("bad shape for the paired product")
    if not gemm_prompt_supported(qtype, k):
        raise ValueError(f"qtype {qtype} / K = {k} has no prompt-sized fused GEMM kernel")
    from .host import device_nbytes
    nb = device_nbytes(qtype, out_features, in_features)
    if w.numel() * w.element_size() != nb or w2.numel() * w2.element_size() != nb:
        raise ValueError("weight buffer size does not match the tiled layout")
    out_dtype = out_dtype or (out.dtype if out is not None else torch.float16)
    if out is None:
        out = torch.empty((m, out_features), dtype=out_dtype, device=x.device)
    if out.shape != (m, out_features) or out.stride(1) != 1 or out.dtype != out_dtype:
        raise ValueError("bad out tensor")
    for b in (bias, bias2):
        if b is not None and (b.dtype != torch.float16 or b.numel() != out_features or not b.is_contiguous()):
            raise ValueError("bias must be contiguous fp16 [out_features]")
    rc = lib().bigdl_gemm_prompt_pair(_ptr(x), _ptr(w), _ptr(w2), _ptr(out), _ptr(bias), _ptr(bias2), m, out_features, k,
                                      x.stride(0), out.stride(0), qtype, ACT[act], _out_type(out_dtype), _stream())
    _check(rc, "gemm_prompt_pair")
    return out


# earlier names of the same entry points
gemm_lowbit_big = gemm_prompt
gemm_lowbit_big_supported = gemm_prompt_supported


def gemm_lowbit_ex(x, w, qtype, out_features, in_features, *, w2=None, bias=None, bias2=None, residual=None, gamma=None,
                   gamma_offset=0.0, eps=1e-6, row_ssq=None, ssq_out=None, act=None, out=None):
    """The MFMA GEMM with the fusions of a batched decode step (17 .. 256 rows):

    * ``gamma`` + ``row_ssq`` [M] fp32: RMSNorm(x) folded in (x is scaled by gamma while it is staged, the result rows by
      ``rsqrt(row_ssq / K + eps)``); ``row_ssq`` holds the sums of squares of the rows of x;
    * ``w2``: gate / up pair in one launch, ``out = act(x W^T + bias) * (x W2^T + bias2)``;
    * ``ssq_out`` [M] fp32 (zeroed by the caller): receives the sums of squares of the rows written to ``out`` (after the
      residual add), i.e. the ``row_ssq`` of the next product."""
    _need_cuda(x, w, w2, bias, bias2, residual, gamma, row_ssq, ssq_out, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D fp16 tensor with contiguous rows")
    m, k = x.shape
    if k != in_features:
        raise ValueError(f"x has {k} features, weight expects {in_features}")
    if qtype not in Q.T16_QTYPES:
        raise ValueError(f"qtype {qtype} has no fused GEMM kernel")
    from .host import device_nbytes
    nb = device_nbytes(qtype, out_features, in_features)
    if w.numel() * w.element_size() != nb or (w2 is not None and w2.numel() * w2.element_size() != nb):
        raise ValueError("weight buffer size does not match the tiled layout")
    if out is None:
        out = torch.empty((m, out_features), dtype=torch.float16, device=x.device)
    if out.shape != (m, out_features) or out.stride(1) != 1:
        raise ValueError("bad out tensor")
    for b_ in (bias, bias2):
        if b_ is not None and (b_.dtype != torch.float16 or b_.numel() != out_features or not b_.is_contiguous()):
            raise ValueError("bias must be contiguous fp16 [out_features]")
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype or residual.stride() != out.stride()):
        raise ValueError("residual must match out")
    if gamma is not None:
        if gamma.dtype != torch.float16 or gamma.numel() != k or not gamma.is_contiguous():
            raise ValueError("gamma must be contiguous fp16 [in_features]")
        if row_ssq is None or row_ssq.dtype != torch.float32 or row_ssq.numel() < m or not row_ssq.is_contiguous():
            raise ValueError("the folded norm needs row_ssq: contiguous fp32 [M]")
    if ssq_out is not None and (ssq_out.dtype != torch.float32 or ssq_out.numel() < m or not ssq_out.is_contiguous()):
        raise ValueError("ssq_out must be contiguous fp32 [M]")
    L = lib()
    sk = L.bigdl_gemm_lowbit_ex_splitk(m, out_features, k, 1 if w2 is not None else 0)
    ws, ws_n = None, 0
    if sk > 1 or w2 is not None or ssq_out is not None:
        ws = _gemm_workspace(x.device, (2 if w2 is not None else 1) * sk * m * out_features)
        ws_n = ws.numel()
    rc = L.bigdl_gemm_lowbit_ex(_ptr(x), _ptr(w), _ptr(w2), _ptr(out), _ptr(bias), _ptr(bias2), _ptr(residual), _ptr(gamma),
                                float(gamma_offset), float(eps), _ptr(row_ssq), _ptr(ssq_out), _ptr(ws), ws_n, m, out_features, k,
                                x.stride(0), out.stride(0), qtype, ACT[act], _out_type(out.dtype), _stream())
    _check(rc, "gemm_lowbit_ex")
    return out


def moe_block_table(counts, bm, max_blocks):
    """counts int64 [E] (rows per expert, rows sorted by expert) -> (blk_expert, blk_row0, blk_rows) int32
    [max_blocks] on the device, built without a host round trip."""
    _need_cuda(counts)
    if counts.dtype != torch.int64 or not counts.is_contiguous():
        raise ValueError("counts must be contiguous int64")
    tbl = torch.empty((3, max_blocks), dtype=torch.int32, device=counts.device)
    rc = lib().bigdl_moe_block_table(_ptr(counts), counts.numel(), bm, max_blocks, _ptr(tbl[0]), _ptr(tbl[1]), _ptr(tbl[2]),
                                     _stream())
    _check(rc, "moe_block_table")
    return tbl[0], tbl[1], tbl[2]


def gemm_lowbit_grouped(x, w, qtype, n_experts, out_features, in_features, table, rows, bm, *, x_rows=None,
                        row_scale=None, act=None, out=None):
    """Grouped GEMM over experts: sorted row i (rows in total, grouped by expert as described by ``table`` =
    ``moe_block_table(...)`` built with the same ``bm``) reads ``x[x_rows[i]]`` (``x[i]`` without x_rows) and
    produces ``out[i] = row_scale[i] * act(x_row @ dequant(W[expert of i])^T)``."""
    blk_expert, blk_row0, blk_rows = table
    _need_cuda(x, w, blk_expert, x_rows, row_scale, out)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1 or x.shape[1] != in_features:
        raise ValueError("x must be a 2-D fp16 tensor [*, in_features] with contiguous rows")
    if qtype not in Q.T16_QTYPES or out_features % 16:
        raise ValueError("grouped GEMM needs a device-tiled format and out_features % 16 == 0")
    from .host import device_nbytes
    if w.numel() * w.element_size() != device_nbytes(qtype, out_features * n_experts, in_features):
        raise ValueError("expert weight buffer size does not match n_experts x [out_features, in_features]")
    max_blocks = blk_expert.numel()
    if max_blocks < (rows + bm - 1) // bm + n_experts - 1 and max_blocks < rows:
        raise ValueError("block table too small for the row count")
    if x_rows is not None and (x_rows.dtype != torch.int32 or x_rows.numel() < rows or not x_rows.is_contiguous()):
        raise ValueError("x_rows must be contiguous int32 [rows]")
    if x_rows is None and x.shape[0] < rows:
        raise ValueError("x has too few rows")
    if row_scale is not None and (row_scale.dtype != torch.float32 or row_scale.numel() < rows or not row_scale.is_contiguous()):
        raise ValueError("row_scale must be contiguous fp32 [rows]")
    if out is None:
        out = torch.empty((rows, out_features), dtype=torch.float16, device=x.device)
    if out.shape != (rows, out_features) or out.stride(1) != 1:
        raise ValueError("bad out tensor")
    rc = lib().bigdl_gemm_lowbit_grouped(_ptr(x), _ptr(w), _ptr(out), _ptr(blk_expert), _ptr(blk_row0), _ptr(blk_rows),
                                         _ptr(x_rows), _ptr(row_scale), rows, max_blocks, bm, n_experts, out_features,
                                         in_features, x.stride(0), out.stride(0), qtype, ACT[act], _out_type(out.dtype),
                                         x.shape[0], _stream())
    _check(rc, "gemm_lowbit_grouped")
    return out


def lora_ops_ok(x, rank) -> bool:
    """whether :func:`lora_down` / :func:`lora_up` take activations like ``x`` ([M, K] rows) with adapters of this rank"""
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.float16, torch.bfloat16) and rank in (8, 16)
            and x.shape[1] % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and 0 < x.shape[0] <= 65535 * 4)


def _lora_small(w, rank, wide):
    """(wt flag, row stride) of an adapter matrix given as [rank, wide] or [wide, rank]"""
    if not w.is_contiguous() or w.data_ptr() % 16:
        raise ValueError("adapter matrices must be contiguous")
    if tuple(w.shape) == (rank, wide) and (rank != wide):
        return 0, wide
    if tuple(w.shape) == (wide, rank):
        return 1, rank
    if tuple(w.shape) == (rank, wide):
        return 0, wide
    raise ValueError(f"adapter matrix {tuple(w.shape)} is neither [{rank}, {wide}] nor [{wide}, {rank}]")


def lora_down(x, w, rank, alpha=1.0):
    """``alpha * x @ W^T`` -> [M, rank] in x's dtype: x [M, K]; w the [rank, K] matrix (an adapter's A) or, as stored,
    its transpose [K, rank] (an adapter's B, for ``gy @ B``).  One pass over x, the small operand in registers."""
    _need_cuda(x, w)
    if not lora_ops_ok(x, rank) or w.dtype != x.dtype:
        raise ValueError("lora_down: fp16 / bf16 rows with K % 8 == 0, rank 8 or 16, adapter in the same dtype")
    m, k = x.shape
    wt, ldw = _lora_small(w, rank, k)
    t = torch.empty((m, rank), dtype=x.dtype, device=x.device)
    rc = lib().bigdl_lora_down(_ptr(x), _ptr(w), _ptr(t), m, k, rank, x.stride(0), ldw, wt, float(alpha), _out_type(x.dtype),
                               _stream())
    _check(rc, "lora_down")
    return t


def lora_up(t, w, out, alpha=1.0, accumulate=True, mask=None):
    """``out = (out if accumulate else 0) + alpha * mask * (t @ W)``: t [M, rank]; w [rank, C] (an adapter's A) or, as
    stored, [C, rank] (an adapter's B, for ``t @ B^T``); out [M, C]; mask a bool / uint8 [M, C] (False = dropped) or None."""
    _need_cuda(t, w, out, mask)
    rank = t.shape[1]
    if not lora_ops_ok(out, rank) or w.dtype != out.dtype or t.dtype != out.dtype or not t.is_contiguous() \
            or t.shape[0] != out.shape[0]:
        raise ValueError("lora_up: fp16 / bf16 rows with C % 8 == 0, rank 8 or 16, operands in one dtype")
    m, c = out.shape
    wt, ldw = _lora_small(w, rank, c)
    ldm = 0
    if mask is not None:
        if mask.dtype not in (torch.bool, torch.uint8) or mask.shape != out.shape or mask.stride(1) != 1 \
                or mask.stride(0) % 8 or mask.data_ptr() % 8:
            raise ValueError("mask must be bool / uint8 [M, C] with 8-byte aligned rows")
        ldm = mask.stride(0)
    rc = lib().bigdl_lora_up(_ptr(t), _ptr(w), _ptr(out), _ptr(mask), m, c, rank, out.stride(0), ldw, ldm, wt, float(alpha),
                             int(bool(accumulate)), _out_type(out.dtype), _stream())
    _check(rc, "lora_up")
    return out


def moe_route(logits, top_k, normalize=True, expert_range=None, sigmoid_scores=False, ids=None, weights=None,
              route_bias=None, n_group=1, topk_group=1, group_top2=False):
    """Router logits [T, E] -> (expert ids int32 [T, top_k], weights fp32 [T, top_k]): softmax, top-k, optional
    renormalisation.  ``expert_range=(lo, hi)`` keeps local experts only (others -1, kept ids rebased to lo).
    ``route_bias`` / ``n_group`` / ``topk_group`` / ``group_top2``: the group-limited selection described at
    :func:`moe_norm_route`."""
    _need_cuda(logits, ids, weights, route_bias)
    if logits.dim() != 2 or logits.stride(1) != 1 or logits.dtype not in (torch.float16, torch.float32):
        raise ValueError("logits must be 2-D fp16/fp32 with contiguous rows")
    t, e = logits.shape
    if ids is None:
        ids = torch.empty((t, top_k), dtype=torch.int32, device=logits.device)
    if weights is None:
        weights = torch.empty((t, top_k), dtype=torch.float32, device=logits.device)
    if ids.numel() != t * top_k or weights.numel() != t * top_k or ids.dtype != torch.int32 \
            or weights.dtype != torch.float32 or not ids.is_contiguous() or not weights.is_contiguous():
        raise ValueError("bad ids / weights buffers")
    lo, hi = expert_range if expert_range is not None else (0, 0)
    if route_bias is not None or n_group > 1:
        if route_bias is not None and (route_bias.dtype != torch.float32 or route_bias.numel() != e
                                       or not route_bias.is_contiguous()):
            raise ValueError("route_bias must be contiguous fp32 [E]")
        if n_group < 1 or n_group > 64 or e % n_group or not 1 <= topk_group <= n_group:
            raise ValueError("bad expert groups")
        rc = lib().bigdl_moe_route_grouped(_ptr(logits), _ptr(ids), _ptr(weights), t, e, top_k, int(bool(normalize)), lo, hi,
                                           logits.stride(0), int(logits.dtype == torch.float32), int(bool(sigmoid_scores)),
                                           _ptr(route_bias), n_group, topk_group, int(bool(group_top2)), _stream())
        _check(rc, "moe_route_grouped")
        return ids, weights
    rc = lib().bigdl_moe_route(_ptr(logits), _ptr(ids), _ptr(weights), t, e, top_k, int(bool(normalize)), lo, hi,
                               logits.stride(0), int(logits.dtype == torch.float32), int(bool(sigmoid_scores)), _stream())
    _check(rc, "moe_route")
    return ids, weights


def moe_norm_route_ok(h, gamma, router, n_experts) -> bool:
    """whether :func:`moe_norm_route` takes these operands"""
    return (h.is_cuda and h.dtype == torch.float16 and h.dim() == 2 and h.stride(1) == 1 and h.stride(0) % 8 == 0
            and h.shape[1] % 8 == 0 and h.shape[1] <= 8192 and h.shape[0] <= 65535
            and gamma.dtype == torch.float16 and gamma.is_contiguous()
            and router.dtype == torch.float16 and router.is_contiguous() and tuple(router.shape) == (n_experts, h.shape[1])
            and n_experts <= 512 and n_experts * h.shape[1] <= 256 * 1024)     # one workgroup reads the whole router


def moe_norm_route(h, gamma, eps, router, top_k, normalize=True, expert_range=None, offset=0.0, sigmoid_scores=False,
                   route_bias=None, n_group=1, topk_group=1, group_top2=False):
    """One launch for the head of a routed layer's decode step: ``x = rmsnorm(h) * (gamma + offset)`` (fp16 [T, H]),
    router logits ``x @ router.T`` (fp32 sums rounded to fp16, like the dense product) and the softmax / top-k of
    :func:`moe_route` -> ``(x, ids int32 [T, top_k], weights fp32 [T, top_k])``.

    ``route_bias`` (fp32 [E]) / ``n_group`` / ``topk_group`` / ``group_top2`` select the way the DeepSeek-V2 / V3 routers
    do: candidates are ``p + bias``, groups of ``E / n_group`` experts are ranked by their best candidate (or the sum of
    their two best), candidates outside the ``topk_group`` best groups become 0, the weights are the picked experts'
    ``p`` (divided by their sum + 1e-20 with ``normalize``)."""
    _need_cuda(h, gamma, router, route_bias)
    e = router.shape[0]
    if route_bias is not None and (route_bias.dtype != torch.float32 or route_bias.numel() != e or not route_bias.is_contiguous()):
        raise ValueError("route_bias must be contiguous fp32 [E]")
    if n_group < 1 or n_group > 64 or e % n_group or not 1 <= topk_group <= n_group:
        raise ValueError("bad expert groups")
    if not moe_norm_route_ok(h, gamma, router, e):
        raise ValueError("moe_norm_route: fp16 rows of H % 8 == 0, H <= 8192 and a contiguous fp16 [E, H] router of at most "
                         "256 Ki elements")
    t, hd = h.shape
    x = torch.empty((t, hd), dtype=torch.float16, device=h.device)
    ids = torch.empty((t, top_k), dtype=torch.int32, device=h.device)
    weights = torch.empty((t, top_k), dtype=torch.float32, device=h.device)
    lo, hi = expert_range if expert_range is not None else (0, 0)
    rc = lib().bigdl_moe_norm_route(_ptr(h), _ptr(gamma), _ptr(router), _ptr(x), _ptr(ids), _ptr(weights), t, hd,
                                    h.stride(0), eps, offset, e, top_k, int(bool(normalize)), lo, hi,
                                    int(bool(sigmoid_scores)), _ptr(route_bias), n_group, topk_group, int(bool(group_top2)),
                                    _stream())
    _check(rc, "moe_norm_route")
    return x, ids, weights


def moe_combine(y, top_k, out, *, shared=None, shared_gate=None, residual=None):
    """``out[t] = residual[t] + sum_k y[t*top_k + k] + sigmoid(shared_gate[t]) * shared[t]`` (fp16 rows)."""
    _need_cuda(y, out, shared, shared_gate, residual)
    t, h = out.shape
    for z in (y, out, shared, residual):
        if z is not None and (z.dtype != torch.float16 or not z.is_contiguous() or z.shape[-1] != h):
            raise ValueError("moe_combine expects contiguous fp16 rows of equal width")
    if y.shape[0] != t * top_k or (shared is not None and shared.shape[0] != t) \
            or (residual is not None and residual.shape != out.shape):
        raise ValueError("moe_combine: row counts do not match")
    ld_gate = 0
    if shared_gate is not None:
        if shared_gate.dtype != torch.float16 or shared_gate.numel() < t:
            raise ValueError("shared_gate must be fp16 with one logit per token")
        ld_gate = shared_gate.stride(0) if shared_gate.dim() > 0 else 1
    rc = lib().bigdl_moe_combine(_ptr(y), _ptr(shared), _ptr(shared_gate), _ptr(residual), _ptr(out), t, top_k, h, ld_gate,
                                 _stream())
    _check(rc, "moe_combine")
    return out


def dequant_t16(w, qtype, out_features, in_features, dtype=torch.float16, out=None, total_rows=None, row0=0):
    """T16 buffer -> dense [out_features, in_features].  With ``total_rows`` the buffer holds a taller
    [total_rows, in_features] matrix and rows [row0, row0 + out_features) are produced (row0 % 16 == 0)."""
    _need_cuda(w, out)
    if out is None:
        out = torch.empty((out_features, in_features), dtype=dtype, device=w.device)
    if out.shape != (out_features, in_features) or out.stride(1) != 1 or out.dtype != dtype:
        raise ValueError("bad out tensor")
    from .host import device_nbytes
    total = total_rows if total_rows is not None else out_features
    if w.numel() * w.element_size() != device_nbytes(qtype, total, in_features):
        raise ValueError("weight buffer size does not match the tiled layout")
    if row0 % 16 or row0 < 0 or row0 + out_features > (total + 15) // 16 * 16:
        raise ValueError("bad row range")
    rc = lib().bigdl_dequant_t16(_ptr(w), _ptr(out), out_features, in_features, out.stride(0), qtype,
                                 _out_type(dtype), total, row0, _stream())
    _check(rc, "dequant_t16")
    return out


def dequant_rows(w, qtype, n_rows, k, ids=None, dtype=torch.float16):
    """host-layout block rows on the device -> dense rows (optionally gathered by ``ids``)."""
    _need_cuda(w, ids)
    from .host import ggml_qk_size, ggml_type_size
    if w.numel() * w.element_size() != n_rows * (k // ggml_qk_size(qtype)) * ggml_type_size(qtype):
        raise ValueError("weight buffer size does not match [n_rows, k]")
    if ids is not None:
        ids = ids.to(torch.int64).contiguous()
        n_out, shape = ids.numel(), (*ids.shape, k)
    else:
        n_out, shape = n_rows, (n_rows, k)
    out = torch.empty(shape, dtype=dtype, device=w.device)
    if n_out == 0:
        return out
    rc = lib().bigdl_dequant_rows(_ptr(w), _ptr(ids), _ptr(out), n_out, n_rows, k, qtype, k, _out_type(dtype),
                                  _stream())
    _check(rc, "dequant_rows")
    return out


def rms_norm(x, weight, eps, *, residual=None, res_out=None, offset=0.0, out=None):
    _need_cuda(x, weight, residual, res_out, out)
    h = x.shape[-1]
    if x.stride(-1) != 1:
        x = x.contiguous()
    x2 = x.reshape(-1, h) if x.is_contiguous() else x
    if x2.dim() != 2:
        x2 = x.contiguous().reshape(-1, h)
    if weight.dtype != x.dtype:
        weight = weight.to(x.dtype)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if residual is not None and not residual.is_contiguous():
        residual = residual.contiguous()
    rc = lib().bigdl_rms_norm(_ptr(x2), _ptr(residual), _ptr(weight.contiguous()), _ptr(out), _ptr(res_out),
                              x2.shape[0], h, x2.stride(0), eps, offset, _out_type(x.dtype), _stream())
    _check(rc, "rms_norm")
    return out


def quant_rows_e4m3(x):
    """x fp16 [M, K] -> (codes ``torch.float8_e4m3fn`` [M, K], scales fp32 [M, 1]) with ``x ~= codes * scales``: the
    row-wise dynamic activation quantization of the opt-in fp8 prompt pass (``torch._scaled_mm`` takes both as they are)."""
    _need_cuda(x)
    if x.dim() != 2 or x.dtype != torch.float16 or x.stride(1) != 1:
        raise ValueError("x must be a 2-D fp16 tensor with contiguous rows")
    m, k = x.shape
    q = torch.empty((m, k), dtype=torch.uint8, device=x.device)
    scale = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    rc = lib().bigdl_quant_rows_e4m3(_ptr(x), _ptr(q), _ptr(scale), m, k, x.stride(0), q.stride(0), _stream())
    _check(rc, "quant_rows_e4m3")
    return q.view(torch.float8_e4m3fn), scale


def bf16_to_f16(x):
    """contiguous bf16 tensor -> fp16 (values beyond the fp16 range saturate)"""
    _need_cuda(x)
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or x.numel() % 8:
        return x.to(torch.float16)
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    _check(lib().bigdl_bf16_to_f16(_ptr(x), _ptr(out), x.numel(), _stream()), "bf16_to_f16")
    return out


def gemv_x_in_lds(m: int, k: int) -> bool:
    """True when ``gemv_lowbit`` stages ``m`` rows of ``k`` features in LDS (the fused attention-combine and LayerNorm
    prologues only exist on that path); mirrors the launcher's budget."""
    xbytes = m * ((k + 127) // 128 * 128 + 136) * 2
    return xbytes <= 48 * 1024 or (m <= 8 and xbytes <= 100 * 1024)


def gemv_lowbit_ln(x, w, qtype, out_features, in_features, gamma, ln, *, w2=None, ln2=None, bias=None, bias2=None,
                   residual=None, eps=1e-5, act=None, out=None):
    """``y = act(LayerNorm(x; gamma, beta) @ dequant(w)^T + bias) [* (LayerNorm(x) @ dequant(w2)^T + bias2)] [+ residual]``
    with the LayerNorm folded into the GEMV: ``ln = (c1, c2)`` are fp32 [N] vectors ``W @ gamma`` and ``W @ beta``
    (``layer_norm_fold``).  fp16 in / out; ``gemv_x_in_lds(M, K)`` must hold."""
    _need_cuda(x, w, w2, bias, bias2, residual, gamma, out, *ln, *(ln2 or ()))
    m, k = x.shape
    if x.dtype != torch.float16 or x.stride(1) != 1 or k != in_features or not gemv_x_in_lds(m, k):
        raise ValueError("x must be fp16 [M, K] with contiguous rows that fit the LDS staging budget")
    if qtype not in Q.T16_QTYPES:
        raise ValueError(f"qtype {qtype} has no fused GEMV kernel")
    if gamma.dtype != torch.float16 or gamma.numel() != k or not gamma.is_contiguous():
        raise ValueError("gamma must be contiguous fp16 [in_features]")
    for pair in (ln, ln2):
        if pair is not None:
            for t in pair:
                if t.dtype != torch.float32 or t.numel() != out_features or not t.is_contiguous():
                    raise ValueError("LayerNorm fold vectors must be contiguous fp32 [out_features]")
    if (w2 is None) != (ln2 is None):
        raise ValueError("w2 and ln2 go together")
    from .host import device_nbytes
    need = device_nbytes(qtype, out_features, in_features)
    for ww in (w, w2):
        if ww is not None and ww.numel() * ww.element_size() != need:
            raise ValueError("weight buffer size does not match [out_features, in_features]")
    for b in (bias, bias2):
        if b is not None and (b.dtype != torch.float16 or b.numel() != out_features or not b.is_contiguous()):
            raise ValueError("bias must be contiguous fp16 [out_features]")
    if out is None:
        out = torch.empty((m, out_features), dtype=torch.float16, device=x.device)
    if out.shape != (m, out_features) or out.stride(1) != 1 or out.dtype not in _DT:
        raise ValueError("bad out tensor")
    if residual is not None and (residual.shape != out.shape or residual.dtype != out.dtype
                                 or residual.stride() != out.stride()):
        raise ValueError("residual must match out")
    rc = lib().bigdl_gemv_lowbit_ln(_ptr(x), _ptr(w), _ptr(w2), _ptr(out), _ptr(bias), _ptr(bias2), _ptr(residual),
                                    _ptr(gamma), eps, m, out_features, k, x.stride(0), out.stride(0), qtype, ACT[act],
                                    _out_type(out.dtype), _ptr(ln[0]), _ptr(ln[1]), _ptr(ln2[0] if ln2 else None),
                                    _ptr(ln2[1] if ln2 else None), _stream())
    _check(rc, "gemv_lowbit_ln")
    return out


def layer_norm_fold(w_dense, gamma, beta):
    """(W @ gamma, W @ beta) in fp32 for ``gemv_lowbit_ln``; ``w_dense`` is the dequantized [N, K] weight, so the
    vectors match what the kernel multiplies with."""
    w = w_dense.float()
    c1 = (w @ gamma.float()).contiguous()
    c2 = (w @ beta.float()).contiguous() if beta is not None else torch.zeros_like(c1)
    return c1, c2


def layer_norm(x, weight, bias, eps):
    _need_cuda(x, weight, bias)
    h = x.shape[-1]
    x2 = x.contiguous().reshape(-1, h)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    w = weight.to(x.dtype).contiguous()
    b = None if bias is None else bias.to(x.dtype).contiguous()
    rc = lib().bigdl_layer_norm(_ptr(x2), _ptr(w), _ptr(b), _ptr(out), x2.shape[0], h, x2.stride(0), eps,
                                _out_type(x.dtype), _stream())
    _check(rc, "layer_norm")
    return out


def act_mul(gate, up, act="silu", out=None):
    """out = act(gate) * up over the last dim; gate/up may be strided row views; out defaults to gate (in place)."""
    _need_cuda(gate, up, out)
    h = gate.shape[-1]
    g2, u2 = gate.reshape(-1, h) if gate.dim() != 2 else gate, up.reshape(-1, h) if up.dim() != 2 else up
    if g2.stride(1) != 1 or u2.stride(1) != 1:
        raise ValueError("last dim must be contiguous")
    o2 = g2 if out is None else (out.reshape(-1, h) if out.dim() != 2 else out)
    rc = lib().bigdl_act_mul(_ptr(g2), _ptr(u2), _ptr(o2), g2.shape[0], h, g2.stride(0), u2.stride(0), o2.stride(0),
                             ACT[act], _out_type(gate.dtype), _stream())
    _check(rc, "act_mul")
    return gate if out is None else out


def silu_mul_bwd(gate, up, grad):
    """(dgate, dup) of ``silu(gate) * up`` for the upstream gradient ``grad`` (same shape; last dim contiguous)."""
    _need_cuda(gate, up, grad)
    h = gate.shape[-1]
    g2, u2, r2 = (t.reshape(-1, h) if t.stride(-1) == 1 else t.contiguous().reshape(-1, h) for t in (gate, up, grad))
    dg = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
    du = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
    rc = lib().bigdl_silu_mul_bwd(_ptr(g2), _ptr(u2), _ptr(r2), _ptr(dg), _ptr(du), g2.shape[0], h, g2.stride(0), u2.stride(0),
                                  r2.stride(0), _out_type(gate.dtype), _stream())
    _check(rc, "silu_mul_bwd")
    return dg, du


def rms_norm_bwd(x, weight, grad, eps, offset=0.0):
    """Gradient of ``rms_norm(x, weight, eps, offset)`` with respect to x."""
    _need_cuda(x, weight, grad)
    h = x.shape[-1]
    x2 = x.reshape(-1, h) if x.stride(-1) == 1 else x.contiguous().reshape(-1, h)
    g2 = grad.reshape(-1, h) if grad.stride(-1) == 1 else grad.contiguous().reshape(-1, h)
    if weight.dtype != x.dtype:
        weight = weight.to(x.dtype)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    rc = lib().bigdl_rms_norm_bwd(_ptr(x2), _ptr(weight.contiguous()), _ptr(g2), _ptr(dx), x2.shape[0], h, x2.stride(0),
                                  g2.stride(0), eps, offset, _out_type(x.dtype), _stream())
    _check(rc, "rms_norm_bwd")
    return dx


def softmax_inplace(x, scale=1.0):
    _need_cuda(x)
    if not x.is_contiguous():
        raise ValueError("softmax_inplace needs a contiguous tensor")
    L = x.shape[-1]
    rc = lib().bigdl_softmax_inplace(_ptr(x), x.numel() // L, L, scale, _out_type(x.dtype), _stream())
    _check(rc, "softmax_inplace")
    return x


def rope_inplace(x, *, position_ids=None, inv_freq=None, cos=None, sin=None, interleaved=False, rot_dim=None,
                 scaling=1.0):
    """In-place rotary embedding on a [B, H, S, D] view (any strides with contiguous last dim)."""
    _need_cuda(x, position_ids, inv_freq, cos, sin)
    if x.dim() != 4 or x.stride(3) != 1:
        raise ValueError("x must be [B, H, S, D] with contiguous last dim")
    b, h, s, d = x.shape
    tab_sb = tab_ld = tab_full = 0
    if inv_freq is not None:
        inv_freq = inv_freq.to(torch.float32).contiguous()
        rot = rot_dim or inv_freq.numel() * 2
        pos = position_ids.to(torch.int64).expand(b, s).contiguous()
    else:
        if cos.dim() == 2:
            cos, sin = cos[None], sin[None]
        cos, sin = cos.to(x.dtype).contiguous(), sin.to(x.dtype).contiguous()
        if cos.shape[-2] != s:
            raise ValueError("cos/sin must have S rows")
        tw = cos.shape[-1]
        rot = rot_dim or (tw if tw <= d and tw * 2 > d else tw * 2 if tw * 2 <= d else tw)
        tab_full = 1 if tw == rot else 0
        tab_ld = tw
        tab_sb = 0 if cos.shape[0] == 1 else s * tw
        pos = None
    rc = lib().bigdl_rope_inplace(_ptr(x), x.stride(0), x.stride(1), x.stride(2), _ptr(pos), _ptr(inv_freq),
                                  _ptr(cos), _ptr(sin), tab_sb, tab_ld, tab_full, b, h, s, d, rot,
                                  int(bool(interleaved)), scaling, _out_type(x.dtype), _stream())
    _check(rc, "rope_inplace")
    return x


def _rows(t, d):
    """[..., d] -> ([rows, d] view with one row stride, True) or (contiguous copy, False) when no such view exists."""
    try:
        return t.view(-1, d), True
    except RuntimeError:
        return t.contiguous().view(-1, d), False


def _last_dim_dense(t):
    return t.shape[-1] == 1 or t.stride(-1) == 1     # (the stride of a one-element dimension is never used)


def to_e5m2(src, dst):
    """src [..., D] float -> dst uint8 of identical logical shape (last dim contiguous on both; slices of a
    larger buffer are fine)."""
    _need_cuda(src, dst)
    if src.shape != dst.shape or dst.dtype != torch.uint8 or not (_last_dim_dense(src) and _last_dim_dense(dst)):
        raise ValueError("to_e5m2: shapes must match, dst uint8, last dim contiguous")
    d = src.shape[-1]
    s2, _ = _rows(src, d)
    d2, direct = _rows(dst, d)
    if not direct:
        d2 = torch.empty(s2.shape, dtype=torch.uint8, device=dst.device)
    rc = lib().bigdl_to_e5m2(_ptr(s2), _ptr(d2), s2.shape[0], d, s2.stride(0), d2.stride(0), _out_type(src.dtype),
                             _stream())
    _check(rc, "to_e5m2")
    if not direct:
        dst.copy_(d2.view(dst.shape))
    return dst


def from_e5m2(src, dst):
    _need_cuda(src, dst)
    if src.shape != dst.shape or src.dtype != torch.uint8 or not (_last_dim_dense(src) and _last_dim_dense(dst)):
        raise ValueError("from_e5m2: shapes must match, src uint8, last dim contiguous")
    d = src.shape[-1]
    s2, _ = _rows(src, d)
    d2, direct = _rows(dst, d)
    if not direct:
        d2 = torch.empty(s2.shape, dtype=dst.dtype, device=dst.device)
    rc = lib().bigdl_from_e5m2(_ptr(s2), _ptr(d2), s2.shape[0], d, s2.stride(0), d2.stride(0), _out_type(dst.dtype),
                               _stream())
    _check(rc, "from_e5m2")
    if not direct:
        dst.copy_(d2.view(dst.shape))
    return dst


def repetition_penalty_inplace(scores, input_ids, penalty):
    _need_cuda(scores, input_ids)
    if not scores.is_contiguous():
        raise ValueError("scores must be contiguous")
    ids = input_ids.to(torch.int64).contiguous()
    b, v = scores.shape
    rc = lib().bigdl_repetition_penalty(_ptr(scores), _ptr(ids), b, ids.shape[1], v, float(penalty),
                                        _out_type(scores.dtype), _stream())
    _check(rc, "repetition_penalty")
    return scores


_ARGMAX_WS = {}


def argmax(x, out=None, inc=None):
    """Row-wise argmax (ties: lowest index).  ``inc=(dst, src, alpha)``: the same launch also does
    ``dst += alpha * src`` on two int32 vectors (the decode graph advances its cache lengths this way)."""
    _need_cuda(x, out)
    v = x.shape[-1]
    x2 = x.reshape(-1, v)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=x.device)
    # long rows of a few-row call are split over several workgroups (candidates + tickets in a per-device workspace)
    nch = 8 if rows <= 64 and v >= 8192 else 1
    ws = None
    if nch > 1:
        key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
        ws = _ARGMAX_WS.get(key)
        if ws is None:
            ws = _ARGMAX_WS[key] = torch.zeros(64 * (2 * nch + 1), dtype=torch.int32, device=x.device)
    inc_dst = inc_src = None
    inc_n = alpha = 0
    if inc is not None:
        inc_dst, inc_src, alpha = inc
        _need_cuda(inc_dst, inc_src)
        if inc_dst.dtype != torch.int32 or inc_src.dtype != torch.int32 or inc_dst.numel() != inc_src.numel() \
                or not (inc_dst.is_contiguous() and inc_src.is_contiguous()):
            raise ValueError("inc: two contiguous int32 vectors of one length")
        inc_n = inc_dst.numel()
    rc = lib().bigdl_argmax(_ptr(x2), _ptr(out), rows, v, x2.stride(0), _out_type(x.dtype), nch, _ptr(ws),
                            _ptr(inc_dst), _ptr(inc_src), inc_n, int(alpha), _stream())
    _check(rc, "argmax")
    return out.reshape(x.shape[:-1])


def attn_decode_ws_floats(b, hq, hkv, q_len, d, splits) -> int:
    return lib().bigdl_attn_decode_ws_floats(b, hq, hkv, q_len, d, splits)


def attn_decode_counters(b, hq, hkv) -> int:
    """int32 arrival counters ``attn_decode`` needs (zeroed once by the caller; every launch leaves them zero)"""
    return lib().bigdl_attn_decode_counters(b, hq, hkv)


def rope_append(qkv, k_cache, v_cache, past_len, inv_freq, *, n_heads, n_kv_heads, head_dim, q_len, batch,
                rot_dim=None, interleaved=False, attn_scaling=1.0, position_ids=None):
    """Rotate q (in place in ``qkv``) and k, and append k / v^T of the ``q_len`` new tokens to the caches."""
    _need_cuda(qkv, k_cache, v_cache, past_len, inv_freq, position_ids)
    if qkv.dtype != torch.float16 or qkv.dim() != 2 or qkv.stride(1) != 1:
        raise ValueError("qkv must be 2-D fp16")
    if qkv.shape[0] != batch * q_len or qkv.shape[1] < (n_heads + 2 * n_kv_heads) * head_dim:
        raise ValueError("qkv shape does not match batch/q_len/heads")
    lmax = k_cache.shape[2]
    if tuple(k_cache.shape) != (batch, n_kv_heads, lmax, head_dim) or tuple(v_cache.shape) != (batch, n_kv_heads, head_dim, lmax):
        raise ValueError("cache shapes must be K [B,Hkv,L,D], V^T [B,Hkv,D,L]")
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()) or k_cache.dtype != v_cache.dtype:
        raise ValueError("caches must be contiguous and of one dtype")
    fp8 = k_cache.dtype == torch.uint8
    if not fp8 and k_cache.dtype != torch.float16:
        raise ValueError("cache dtype must be float16 or uint8 (e5m2)")
    if past_len.dtype != torch.int32 or past_len.numel() != batch:
        raise ValueError("past_len must be int32 [batch]")
    if inv_freq is not None:
        inv_freq = inv_freq.to(torch.float32).contiguous()
        rot_dim = rot_dim or inv_freq.numel() * 2
    if position_ids is not None and (position_ids.dtype != torch.int32 or position_ids.numel() != batch * q_len):
        raise ValueError("position_ids must be int32 [batch*q_len]")
    rc = lib().bigdl_rope_append(_ptr(qkv), _ptr(k_cache), _ptr(v_cache), _ptr(past_len), _ptr(inv_freq),
                                 _ptr(position_ids), attn_scaling, batch, n_heads, n_kv_heads, q_len, qkv.stride(0),
                                 lmax, head_dim, rot_dim or 0, int(bool(interleaved)), int(fp8), _stream())
    _check(rc, "rope_append")


def attn_decode(qkv, k_cache, v_cache, past_len, out, ws, counters, *, n_heads, n_kv_heads, head_dim, q_len, batch,
                scale, splits, softcap=0.0, window=0, mask=None, rope_table=None, rot_dim=None, interleaved=False,
                position_ids=None, append=False, defer_combine=False, alibi=None, k_scale=None, v_scale=None):
    """Split-KV decode attention over the fixed-capacity caches.

    ``k_scale`` / ``v_scale``: int4 caches (``kv_int4_shapes``): K [B,Hkv,L,D/2] and V^T [B,Hkv,D/2,L] bytes with fp16
    scales per key and 32 channels; the kernel always appends itself in that mode (``rope_table`` or ``append=True``).

    ``alibi``: optional fp32 [n_heads] slopes; the score of key position j for the query at position i gets the ALiBi
    bias ``slope * (j - i)`` added.

    With ``rope_table`` ([max_pos, rot/2, 2] fp32 cos/sin) the kernel also rotates q in registers and
    rotates + appends the new k / v^T (one launch instead of rope_append + attention); ``append=True``
    without a table appends unrotated k/v.  Row counts above 32 are processed in token chunks.
    """
    _need_cuda(qkv, k_cache, v_cache, past_len, out, ws, counters, mask, rope_table, position_ids, k_scale, v_scale)
    lmax = k_cache.shape[2]
    if qkv.dtype != torch.float16 or out.dtype != torch.float16 or qkv.dim() != 2 or qkv.stride(1) != 1:
        raise ValueError("qkv/out must be fp16, qkv 2-D with contiguous rows")
    if qkv.shape[0] != batch * q_len or qkv.shape[1] < (n_heads + 2 * n_kv_heads) * head_dim:
        raise ValueError("qkv shape does not match batch/q_len/heads")
    if tuple(out.shape) != (batch * q_len, n_heads * head_dim) or not out.is_contiguous():
        raise ValueError("out must be contiguous [batch*q_len, n_heads*head_dim]")
    int4 = k_scale is not None
    if int4:
        _check_kv_int4(k_cache, v_cache, k_scale, v_scale, batch, n_kv_heads, head_dim)
        if rope_table is None and not append:
            raise ValueError("int4 caches are appended by the attention kernel (rope_table or append=True)")
    elif tuple(k_cache.shape) != (batch, n_kv_heads, lmax, head_dim) or tuple(v_cache.shape) != (batch, n_kv_heads, head_dim, lmax):
        raise ValueError("cache shapes must be K [B,Hkv,L,D], V^T [B,Hkv,D,L]")
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()) or k_cache.dtype != v_cache.dtype:
        raise ValueError("caches must be contiguous and of one dtype")
    if lmax % 256:
        raise ValueError("cache capacity must be a multiple of 256")
    fp8 = k_cache.dtype == torch.uint8 and not int4
    if not fp8 and not int4 and k_cache.dtype != torch.float16:
        raise ValueError("cache dtype must be float16 or uint8 (e5m2)")
    g = n_heads // n_kv_heads
    nchunk = (g + 31) // 32                      # groups above 32 query heads are split over workgroup columns
    gchunk = (g + nchunk - 1) // nchunk
    chunk = max(1, 32 // gchunk)
    need = attn_decode_ws_floats(batch, n_heads, n_kv_heads, min(q_len, chunk), head_dim, splits)
    if ws.dtype != torch.float32 or ws.numel() < need:
        raise ValueError(f"workspace needs {need} floats")
    if counters.dtype != torch.int32 or counters.numel() < batch * n_kv_heads * nchunk:
        raise ValueError("counters must be int32 [attn_decode_counters(batch, n_heads, n_kv_heads)]")
    if defer_combine and nchunk > 1:
        raise ValueError("defer_combine needs at most 32 query heads per kv head")
    if past_len.dtype != torch.int32 or past_len.numel() != batch:
        raise ValueError("past_len must be int32 [batch]")
    mask_ld = 0
    if mask is not None:
        if mask.dtype != torch.float16 or not mask.is_contiguous() or mask.shape[0] != batch or mask.shape[-2] != q_len:
            raise ValueError("mask must be contiguous fp16 [B, q_len, L]")
        mask_ld = mask.shape[-1]
    rope_max = rot = 0
    if rope_table is not None:
        if rope_table.dtype != torch.float32 or rope_table.dim() != 3 or rope_table.shape[2] != 2 or not rope_table.is_contiguous():
            raise ValueError("rope_table must be contiguous fp32 [max_pos, rot/2, 2]")
        rope_max, rot = rope_table.shape[0], rope_table.shape[1] * 2
        if rot_dim is not None and rot_dim != rot:
            raise ValueError("rot_dim does not match the rope table")
        if not interleaved and rot != head_dim:
            raise ValueError("fused rotate-half RoPE needs rot_dim == head_dim (use rope_append otherwise)")
    if position_ids is not None and (position_ids.dtype != torch.int32 or position_ids.numel() != batch * q_len):
        raise ValueError("position_ids must be int32 [batch*q_len]")
    if defer_combine and q_len > chunk:
        raise ValueError("defer_combine needs all rows in one launch")
    _check_alibi(alibi, n_heads)
    for off in range(0, q_len, chunk):
        cnt = min(chunk, q_len - off)
        do_append = (rope_table is not None or append) and off == 0
        rc = lib().bigdl_attn_decode(_ptr(qkv), _ptr(k_cache), _ptr(v_cache), _ptr(out), _ptr(ws), _ptr(counters),
                                     _ptr(past_len), _ptr(mask), mask_ld, _ptr(alibi),
                                     _ptr(rope_table), _ptr(position_ids), rope_max, rot,
                                     int(bool(interleaved)), scale, softcap, batch, n_heads, n_kv_heads, q_len, off, cnt,
                                     qkv.stride(0), lmax, head_dim, splits, window, int(fp8), int(bool(do_append)),
                                     int(bool(defer_combine)), _ptr(k_scale), _ptr(v_scale), _stream())
        _check(rc, "attn_decode")
    return out


KV_INT4_HEAD_DIMS = (64, 128, 256)


def kv_int4_shapes(batch, n_kv_heads, capacity, head_dim):
    """Shapes of the four tensors of one layer's int4 cache: K bytes, V^T bytes (uint8), K scales, V scales (fp16).
    K: one 32-bit word per 8 channels of a key, channel j in nibble (j >> 1) + 4 (j & 1); V^T row 16m + r holds channels
    32m + r (low nibble) and 32m + 16 + r (high nibble); value = (nibble - 8) * scale, one scale per key and 32 channels."""
    d = head_dim
    return ((batch, n_kv_heads, capacity, d // 2), (batch, n_kv_heads, d // 2, capacity),
            (batch, n_kv_heads, capacity, d // 32), (batch, n_kv_heads, d // 32, capacity))


def _check_kv_int4(k_cache, v_cache, k_scale, v_scale, batch, n_kv_heads, head_dim):
    if head_dim not in KV_INT4_HEAD_DIMS:
        raise ValueError(f"int4 caches need head_dim in {KV_INT4_HEAD_DIMS}")
    if v_scale is None or k_scale is None:
        raise ValueError("int4 caches need both scale tensors")
    want = kv_int4_shapes(batch, n_kv_heads, k_cache.shape[2], head_dim)
    got = tuple(tuple(t.shape) for t in (k_cache, v_cache, k_scale, v_scale))
    if got != want:
        raise ValueError(f"int4 cache shapes {got} != {want}")
    if k_cache.dtype != torch.uint8 or v_cache.dtype != torch.uint8 or k_scale.dtype != torch.float16 or v_scale.dtype != torch.float16:
        raise ValueError("int4 caches are uint8 with fp16 scales")
    if not all(t.is_contiguous() for t in (k_cache, v_cache, k_scale, v_scale)):
        raise ValueError("int4 caches must be contiguous")


def kv_int4_append(qkv, k_cache, v_cache, k_scale, v_scale, *, n_heads, n_kv_heads, head_dim, past):
    """Quantize the (already rotated) keys and the values of a prompt, qkv [B, S, (Hq + 2 Hkv) D] fp16, into the int4
    caches at positions past .. past + S."""
    _need_cuda(qkv, k_cache, v_cache, k_scale, v_scale)
    if qkv.dtype != torch.float16 or qkv.dim() != 3 or qkv.stride(2) != 1 or qkv.shape[2] < (n_heads + 2 * n_kv_heads) * head_dim:
        raise ValueError("qkv must be fp16 [B, S, (Hq + 2 Hkv) D]")
    b, s = qkv.shape[:2]
    if b > 1 and qkv.stride(0) != s * qkv.stride(1):
        raise ValueError("qkv batch stride must be S rows")
    _check_kv_int4(k_cache, v_cache, k_scale, v_scale, b, n_kv_heads, head_dim)
    if past < 0 or past + s > k_cache.shape[2]:
        raise ValueError("KV cache capacity exceeded")
    rc = lib().bigdl_kv_int4_append(_ptr(qkv), _ptr(k_cache), _ptr(v_cache), _ptr(k_scale), _ptr(v_scale), b, n_heads,
                                    n_kv_heads, s, head_dim, past, k_cache.shape[2], qkv.stride(1), _stream())
    _check(rc, "kv_int4_append")


def kv_int4_dequant(k_cache, v_cache, k_scale, v_scale, length):
    """fp16 K [B,Hkv,length,D] and V^T [B,Hkv,D,length] of the first ``length`` positions of an int4 cache (prompt
    passes over a non-empty cache, tests); plain tensor ops."""
    b, h, _, d2 = k_cache.shape
    d = 2 * d2
    kb = k_cache[:, :, :length].reshape(b, h, length, d // 8, 4).to(torch.int32)
    w = kb[..., 0] | (kb[..., 1] << 8) | (kb[..., 2] << 16) | (kb[..., 3] << 24)                  # one word per 8 channels
    j = torch.arange(8, device=k_cache.device)
    nib = (w[..., None] >> (4 * ((j >> 1) + 4 * (j & 1)))) & 15                                  # [.., D/8, 8]
    k = (nib - 8).to(torch.float16).reshape(b, h, length, d // 32, 32) * k_scale[:, :, :length, :, None]
    vb = v_cache[:, :, :, :length].reshape(b, h, d // 32, 16, length).to(torch.int32)
    lo, hi = (vb & 15) - 8, (vb >> 4) - 8                                                         # channels 32m+r, 32m+16+r
    v = torch.cat([lo, hi], 3).to(torch.float16) * v_scale[:, :, :, None, :length]                # [B,H,D/32,32,L]
    return k.reshape(b, h, length, d), v.reshape(b, h, d, length)


PREFILL_HEAD_DIMS = (64, 72, 80, 96, 128, 256)


def _check_alibi(alibi, n_heads):
    if alibi is not None and (alibi.dtype != torch.float32 or alibi.numel() != n_heads or not alibi.is_contiguous()
                              or alibi.device.type != "cuda"):
        raise ValueError("alibi must be a contiguous fp32 device tensor [n_heads]")


def attn_prefill(q, k, v, *, scale, causal=True, mask=None, softcap=0.0, window=0, out=None, v_transposed=False,
                 alibi=None, kv_e5m2=False):
    """Flash-style attention for many query tokens.  q [B, Hq, Sq, D], k / v [B, Hkv, Lk, D]: fp16, any strides with
    a contiguous last dim (views into a fused qkv buffer or a KV cache are fine); with ``v_transposed`` v is
    [B, Hkv, D, Lk] (the engine's V^T cache layout, row stride multiple of 8).  mask: optional additive fp16
    [B, 1 | Hq | none, Sq, Lk].  alibi: optional fp32 [Hq] slopes (score += slope * (key - query position)).  Returns out [B, Sq, Hq * D].  Causal alignment: query i sees keys <= i + (Lk - Sq)."""
    _need_cuda(q, k, v, mask, out)
    b, hq, sq, d = q.shape
    hkv, lk = k.shape[1], k.shape[2]
    if d not in PREFILL_HEAD_DIMS:
        raise ValueError("attn_prefill supports head_dim 64, 72, 80, 96, 128 and 256")
    if kv_e5m2 and (not v_transposed or k.dtype != torch.uint8 or v.dtype != torch.uint8 or d in (72, 80)):
        raise ValueError("kv_e5m2 takes uint8 K [B, Hkv, Lk, D] and V^T [B, Hkv, D, Lk] (the engine's fp8 cache layout)")
    for t in (q,) if kv_e5m2 else (q, k, v):
        if t.dtype != torch.float16 or t.stride(-1) != 1:
            raise ValueError("q / k / v must be fp16 with a contiguous last dim")
    if kv_e5m2 and (k.stride(-1) != 1 or v.stride(-1) != 1):
        raise ValueError("cache views must have a contiguous last dim")
    vshape = (b, hkv, d, lk) if v_transposed else (b, hkv, lk, d)
    if tuple(v.shape) != vshape or k.shape[0] != b or k.shape[3] != d or hq % hkv:
        raise ValueError("inconsistent q / k / v shapes")
    if v_transposed and (v.stride(2) % 8 or v.data_ptr() % 16):
        raise ValueError("transposed v needs 16-byte aligned rows")
    if causal and lk < sq:
        raise ValueError("causal attention needs at least as many keys as queries")
    if out is None:
        out = torch.empty((b, sq, hq * d), dtype=torch.float16, device=q.device)
    if out.shape != (b, sq, hq * d) or out.stride(-1) != 1 or out.dtype != torch.float16:
        raise ValueError("bad out tensor")
    m_sb = m_sq = m_sh = 0
    if mask is not None:
        if mask.dim() == 3:
            mask = mask[:, None]
        if mask.dtype != torch.float16 or mask.dim() != 4 or mask.stride(-1) != 1 or mask.shape[-2:] != (sq, lk) \
                or mask.shape[0] not in (1, b) or mask.shape[1] not in (1, hq):
            raise ValueError("mask must be additive fp16 [B, (1 | Hq,) Sq, Lk] with a contiguous last dim")
        m_sb = mask.stride(0) if mask.shape[0] == b else 0
        m_sh = mask.stride(1) if mask.shape[1] == hq and hq > 1 else 0
        m_sq = mask.stride(2)
    _check_alibi(alibi, hq)
    rc = lib().bigdl_attn_prefill(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(mask), b, hq, hkv, sq, lk, d,
                                  q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                  v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), m_sb, m_sq, m_sh,
                                  _ptr(alibi), float(scale), float(softcap or 0.0), int(bool(causal)), int(window or 0),
                                  int(bool(v_transposed)), int(bool(kv_e5m2)), _stream())
    _check(rc, "attn_prefill")
    return out


def prefill_rope_append(qkv, k_cache, v_cache, rope_table, *, n_heads, n_kv_heads, head_dim, past, rot_dim=None,
                        interleaved=False, position_ids=None, vt_scratch=None):
    """One launch per layer for a prompt: rotate q and k in place in ``qkv`` [B, S, (Hq+2Hkv)*D], copy k to the K
    cache [B, Hkv, Lmax, D] and v transposed to the V^T cache [B, Hkv, D, Lmax] (fp16 or e5m2 by the cache dtype);
    ``vt_scratch`` [B, Hkv, D, S_alloc] fp16 additionally receives V^T unquantized (for e5m2 caches).  Without caches
    (both None) only the rotation and the scratch copy happen."""
    _need_cuda(qkv, k_cache, v_cache, rope_table, position_ids, vt_scratch)
    b, s, width = qkv.shape
    d = head_dim
    if qkv.dtype != torch.float16 or qkv.stride(2) != 1 or qkv.stride(0) != s * qkv.stride(1) or width != (n_heads + 2 * n_kv_heads) * d:
        raise ValueError("qkv must be fp16 [B, S, (Hq+2Hkv)*D] with uniform row stride")
    if k_cache is None:
        # rotation only (int4 caches are filled by kv_int4_append from the rotated qkv); V^T goes to the scratch
        if v_cache is not None:
            raise ValueError("pass both caches or none")
        fp8, lmax = True, past + s
    else:
        fp8 = k_cache.dtype == torch.uint8
        lmax = k_cache.shape[2]
        if tuple(k_cache.shape) != (b, n_kv_heads, lmax, d) or tuple(v_cache.shape) != (b, n_kv_heads, d, lmax) \
                or not k_cache.is_contiguous() or not v_cache.is_contiguous() or v_cache.dtype != k_cache.dtype:
            raise ValueError("cache shapes do not match")
        if past + s > lmax:
            raise ValueError("KV cache capacity exceeded")
    rot = d if rot_dim is None else rot_dim
    if rope_table.dtype != torch.float32 or not rope_table.is_contiguous() or (rot and rope_table.shape[1] != rot // 2):
        raise ValueError("rope table must be fp32 [max_pos, rot/2, 2]")
    if position_ids is not None:
        position_ids = position_ids.to(torch.int64).contiguous()
        if position_ids.numel() != b * s:
            raise ValueError("position_ids must be [B, S]")
    vt_ld = 0
    if vt_scratch is not None:
        if vt_scratch.dtype != torch.float16 or vt_scratch.shape[:3] != (b, n_kv_heads, d) or vt_scratch.shape[3] < s \
                or not vt_scratch.is_contiguous():
            raise ValueError("vt_scratch must be contiguous fp16 [B, Hkv, D, >= S]")
        vt_ld = vt_scratch.shape[3]
    rc = lib().bigdl_prefill_rope_append(_ptr(qkv), _ptr(k_cache), _ptr(v_cache), _ptr(vt_scratch), _ptr(rope_table),
                                         _ptr(position_ids), b, n_heads, n_kv_heads, s, d, past, lmax, rot,
                                         int(bool(interleaved)), int(fp8), qkv.stride(1), vt_ld, rope_table.shape[0], _stream())
    _check(rc, "prefill_rope_append")


def rwkv_time_shift(hidden, shifted, mix):
    """hidden / shifted [B, T, C], mix [n, C] -> [n, B, T, C]: hidden * mix[i] + shifted * (1 - mix[i])."""
    _need_cuda(hidden, shifted, mix)
    hidden, shifted = hidden.contiguous(), shifted.contiguous().view_as(hidden)
    mix = mix.reshape(-1, hidden.shape[-1]).to(hidden.dtype).contiguous()
    out = torch.empty((mix.shape[0], *hidden.shape), dtype=hidden.dtype, device=hidden.device)
    rc = lib().bigdl_rwkv_time_shift(_ptr(hidden), _ptr(shifted), _ptr(mix), _ptr(out), hidden.numel() // hidden.shape[-1],
                                     hidden.shape[-1], mix.shape[0], _out_type(hidden.dtype), _stream())
    _check(rc, "rwkv_time_shift")
    return out


def rwkv_linear_attention_v4(time_decay, time_first, key, value, num_state=None, den_state=None, max_state=None):
    """RWKV-4 wkv.  time_decay is the raw parameter (the kernel uses -exp(time_decay)); key / value [B, T, C];
    the three fp32 [B, C] states are read (when given) and updated in place."""
    _need_cuda(time_decay, time_first, key, value, num_state, den_state, max_state)
    b, t, c = key.shape
    key, value = key.contiguous(), value.to(key.dtype).contiguous()
    w = (-torch.exp(time_decay.float())).contiguous()
    u = time_first.float().contiguous()
    out = torch.empty_like(key)
    has = num_state is not None
    for s_ in (num_state, den_state, max_state):
        if s_ is not None and (s_.dtype != torch.float32 or not s_.is_contiguous() or s_.shape != (b, c)):
            raise ValueError("wkv states must be contiguous fp32 [B, C]")
    rc = lib().bigdl_rwkv4_wkv(_ptr(w), _ptr(u), _ptr(key), _ptr(value), _ptr(out), _ptr(num_state), _ptr(den_state),
                               _ptr(max_state), b, t, c, _out_type(key.dtype), int(has), _stream())
    _check(rc, "rwkv4_wkv")
    return out


def rwkv_linear_attention_v5(time_decay, time_first, receptance, key, value, state):
    """RWKV-5 wkv.  time_decay (already exp(-exp(w))) / time_first [H, S]; r / k / v [B, H, T, S] fp32;
    state [B, H, S, S] fp32 updated in place.  Returns [B, T, H, S] fp32."""
    _need_cuda(time_decay, time_first, receptance, key, value, state)
    b, h, t, s_ = key.shape
    args = [x.float().contiguous() for x in (time_decay.reshape(h, s_), time_first.reshape(h, s_), receptance, key, value)]
    if state.dtype != torch.float32 or not state.is_contiguous() or state.shape != (b, h, s_, s_):
        raise ValueError("state must be contiguous fp32 [B, H, S, S]")
    out = torch.empty((b, t, h, s_), dtype=torch.float32, device=key.device)
    rc = lib().bigdl_rwkv5_wkv(*map(_ptr, args), _ptr(state), _ptr(out), b, h, t, s_, _stream())
    _check(rc, "rwkv5_wkv")
    return out
