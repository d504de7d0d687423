"""Quantized parameter + linear modules.

Capability map to the reference (python/llm/src/ipex_llm/transformers/low_bit_linear.py):
``FP4Params`` :312-459, ``ggml_convert_qtype`` :104-175, ``MatMulLowBit`` :500-541,
``LowBitLinear`` :573-715, ``FP16Linear``/``BF16Linear`` :718-873, vLLM shims :876-908.

MI355X design differences:
* one device code path (HIP, gfx950): the fused MFMA GEMV serves 1..64 rows, larger row counts
  dequantize tile-wise to fp16 and use the hipBLASLt GEMM; there is no per-device heuristic table.
* the host keeps ggml block order (what checkpoints store); ``.to('cuda')`` re-tiles to the
  "T16" kernel layout on the host and does a single H2D copy; ``.to('cpu')`` reverses it.
* tensor-parallel all-reduce goes through ``torch.distributed`` (RCCL) on ``mp_group``.
"""
from __future__ import annotations

import os
from functools import reduce
from operator import mul

import torch
import torch.nn.functional as F
from torch import nn

from .. import qtypes as Q
from ..ops import host
from ..utils.common import invalidInputError

TORCH_LINEAR_THRESHOLD = int(os.getenv("BIGDL_LLM_LINEAR_THRESHOLD", "512"))
# rows above which the device path dequantizes + calls the dense GEMM instead of the fused GEMV
# row-count policy of the fused kernels on the GPU (measured on the Llama-2-7B projection shapes, sym_int4): the skinny
# GEMV wins up to 16 rows (one row tile), the fused MFMA GEMM from 17 to 256 rows (in isolation the two-tile GEMV ties it
# up to 32 rows, inside the batched decode step the GEMM path is 20 % ahead at 32 rows), dequantize + hipBLASLt above
GEMV_MAX_ROWS = int(os.getenv("BIGDL_GEMV_MAX_ROWS", "16"))
GEMM_MAX_ROWS = 256
# forward of frozen low-bit linears inside a training step on the fused prompt GEMM (fp16 copy of the activations)
# instead of dequantize + dense product
_TRAIN_FUSED_FORWARD = os.getenv("BIGDL_TRAIN_FUSED_FORWARD", "0") == "1"


def get_block_size(qtype) -> int:
    return host.ggml_qk_size(Q.qtype_id(qtype))


def get_qk_size(qtype: int) -> int:
    return host.ggml_qk_size(qtype)


def ggml_convert_qtype(tensor: torch.Tensor, qtype: int, device=None, convert_shape_only=False,
                       imatrix: torch.Tensor = None, in_features: int = None, enable_scale_search: bool = False):
    invalidInputError(tensor.dtype == torch.float, "Input tensor must be float32")
    return host.quantize(tensor, qtype, imatrix=imatrix if qtype in Q.K_QTYPES or qtype in Q.RTN_QTYPES else None,
                         scale_search=enable_scale_search, shape_only=convert_shape_only, device=device)


def ggml_convert_fp32(tensor: torch.Tensor, weight_shape, k: int, qtype: int):
    raw = tensor.view(torch.uint8) if tensor.dtype != torch.uint8 else tensor
    return host.dequantize(raw, qtype, weight_shape)


def ggml_int4_convert_fp32(tensor: torch.Tensor, weight_shape, k: int):
    return ggml_convert_fp32(tensor, weight_shape, k, Q.SYM_INT4)


def ggml_q_format_convet_cpu2xpu(tensor: torch.Tensor, shape, qtype: int):
    """host block order -> device tile order (name kept from the reference API)."""
    if qtype not in Q.T16_QTYPES:
        return tensor
    raw = tensor.view(torch.uint8) if tensor.dtype != torch.uint8 else tensor
    return host.relayout_to_device(raw, qtype, shape[0], shape[1])


def ggml_q_format_convet_xpu2cpu(tensor: torch.Tensor, shape, qtype: int):
    if qtype not in Q.T16_QTYPES:
        return tensor
    return host.relayout_to_host(tensor, qtype, shape[0], shape[1])


class FP4Params(torch.nn.Parameter):
    """A frozen block-quantized weight.

    ``data`` is fp32/fp16 before quantization, afterwards a flat byte buffer:
    host layout on CPU/meta, T16 tile layout on the HIP device (``layout`` says which).
    """

    def __new__(cls, data=None, requires_grad=False, quantized=False, _shape=None, convert_shape_only=False,
                qtype=None, imatrix=None, in_features=None, enable_scale_search=False, qlayout=None,
                host_only=False):
        if data is None:
            data = torch.empty(0)
        self = torch.Tensor._make_subclass(cls, data, requires_grad)
        self.data = data
        self.quantized = quantized
        self._shape = _shape
        self.qtype = qtype
        self.convert_shape_only = convert_shape_only
        self.imatrix = imatrix
        self.in_features = in_features
        self.enable_scale_search = enable_scale_search
        # rows gathered one by one (embeddings) keep the host block order even on the device
        self.host_only = host_only or (qlayout == "host" and data.device.type == "cuda")
        return self

    @property
    def qlayout(self) -> str:
        """"t16" (device tile order) or "host" (ggml block order).  Derived from where the bytes live, because
        ``nn.Module._apply`` swaps ``.data`` in place and drops attributes of the tensor ``to()`` returned."""
        if self.quantized and self.data.device.type == "cuda" and self.qtype in Q.T16_QTYPES and not self.host_only:
            return "t16"
        return "host"

    def _clone_with(self, data, layout=None):
        return FP4Params(data, requires_grad=self.requires_grad, quantized=self.quantized, _shape=self._shape,
                         qtype=self.qtype, enable_scale_search=self.enable_scale_search,
                         host_only=self.host_only)

    def __deepcopy__(self, memo):
        if id(self) in memo:
            return memo[id(self)]
        new = FP4Params(self.data.clone(), requires_grad=self.requires_grad, quantized=self.quantized,
                        _shape=self._shape, convert_shape_only=self.convert_shape_only, qtype=self.qtype,
                        imatrix=self.imatrix, in_features=self.in_features,
                        enable_scale_search=self.enable_scale_search, host_only=self.host_only)
        if getattr(self, "_host_blocks", None) is not None:
            new._host_blocks = self._host_blocks
        memo[id(self)] = new
        return new

    def ggml_mse(self, w, ggml_qtype, device):
        w_quant = ggml_convert_qtype(w, ggml_qtype, device=device, convert_shape_only=self.convert_shape_only,
                                     enable_scale_search=self.enable_scale_search)
        w_dequant = ggml_convert_fp32(w_quant, w.shape, reduce(mul, w.shape, 1), ggml_qtype)
        return F.mse_loss(w_dequant, w), w_quant

    def quantize(self, device=None):
        if not self.quantized:
            w = self.data.contiguous().float()
            shape_only = self.convert_shape_only or device == "meta"
            if self.qtype in (Q.MOFQ4, Q.MOFQ8):
                # mixture of formats: keep whichever of the int / fp format reconstructs this tensor better
                a, b = (Q.SYM_INT4, Q.FP4) if self.qtype == Q.MOFQ4 else (Q.SYM_INT8, Q.FP8E4)
                if shape_only:
                    self.data = ggml_convert_qtype(w, a, device=device, convert_shape_only=True)
                    self.qtype = a
                else:
                    mse_a, q_a = self.ggml_mse(w, a, device)
                    mse_b, q_b = self.ggml_mse(w, b, device)
                    self.qtype, self.data = (a, q_a) if mse_a <= mse_b else (b, q_b)
            else:
                self.data = ggml_convert_qtype(w, self.qtype, device=device, convert_shape_only=shape_only,
                                               imatrix=self.imatrix, in_features=self.in_features,
                                               enable_scale_search=self.enable_scale_search)
            self.quantized = True
            self._shape = w.shape
        return self

    def get_shape(self):
        return self._shape

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        src = self.data.device.type
        if device is None:
            return self._clone_with(self.data) if dtype is None else self  # dtype casts never touch the codes
        dst = device.type
        if dst == "cpu" and src == "cpu":
            return self.quantize("cpu")
        if dst == "meta" and src == "meta":
            return self.quantize("meta")
        if dst == "cuda" and src == "cpu":
            self.quantize("cpu")
            if self.qtype in Q.T16_QTYPES and not self.host_only:
                tiled = ggml_q_format_convet_cpu2xpu(self.data, self._shape, self.qtype)
                new = self._clone_with(tiled.to(device, non_blocking=non_blocking), layout="t16")
                if self.qtype in Q.T16_ONE_WAY:
                    # k-quant sub-scales are folded into per-32 fp16 scales on the device: keep the host blocks for
                    # save_low_bit / .to("cpu").  nn.Module._apply keeps this object and swaps its .data, a direct
                    # caller keeps the returned one: both carry the copy.
                    self._host_blocks = new._host_blocks = self.data
                return new
            return self._clone_with(self.data.to(device, non_blocking=non_blocking), layout="host")
        if dst == "cpu" and src == "cuda":
            if self.qlayout == "t16" and self.qtype in Q.T16_ONE_WAY:
                blocks = getattr(self, "_host_blocks", None)
                if blocks is None:
                    raise RuntimeError("this k-quant weight was re-sliced on the device; its host blocks are gone")
                return self._clone_with(blocks, layout="host")
            data = self.data.to("cpu")
            if self.qlayout == "t16":
                data = ggml_q_format_convet_xpu2cpu(data, self._shape, self.qtype)
            elif self.qtype == Q.NF4 and data.dtype == torch.uint8:
                data = data.view(torch.bfloat16)
            return self._clone_with(data, layout="host")
        return self._clone_with(self.data.to(device, non_blocking=non_blocking))

    # nn.Module._apply calls these when moving modules between devices
    def cuda(self, device=None, non_blocking=False):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device),
                       non_blocking=non_blocking)

    def cpu(self):
        return self.to("cpu")

    # dtype casts (model.half(), .bfloat16(), .float()) must never reinterpret quantized bytes
    def half(self, *a, **k):
        return self if self.quantized else super().half(*a, **k)

    def bfloat16(self, *a, **k):
        return self if self.quantized else super().bfloat16(*a, **k)

    def float(self, *a, **k):
        return self if self.quantized else super().float(*a, **k)


# ---------------------------------------------------------------------------
# functional kernels (device dispatch lives here, nowhere else)
# ---------------------------------------------------------------------------
_TRAIN_DENSE_CACHE = os.environ.get("BIGDL_TRAIN_DENSE_CACHE") == "1"


def train_dense_cache_enabled() -> bool:
    return _TRAIN_DENSE_CACHE


def auto_train_dense_cache(model, compute_dtype=torch.bfloat16) -> bool:
    """Called when a low-bit model is prepared for training.  A QLoRA step expands every frozen weight twice (forward
    and backward: 8 of 64 ms on Llama-2-7B).  On a 288 GB device the dense copies of a 7B - 13B model are a few percent
    of the memory, so they are kept next to the low-bit weights when they fit comfortably: at most an eighth of the
    device and a third of what is free now.  ``BIGDL_TRAIN_DENSE_CACHE=0`` / ``=1`` decide by hand."""
    global _TRAIN_DENSE_CACHE
    env = os.environ.get("BIGDL_TRAIN_DENSE_CACHE")
    if env is not None:
        _TRAIN_DENSE_CACHE = env == "1"
        return _TRAIN_DENSE_CACHE
    dense, dev = 0, None
    for p in model.parameters():
        if isinstance(p, FP4Params) and getattr(p, "quantized", False) and p.data.device.type == "cuda" \
                and getattr(p, "qlayout", None) == "t16":
            n, k = p._shape
            dense += n * k * torch.empty((), dtype=compute_dtype).element_size()
            dev = p.data.device
    if dev is None:
        return _TRAIN_DENSE_CACHE
    free, total = torch.cuda.mem_get_info(dev)
    _TRAIN_DENSE_CACHE = dense <= total // 8 and dense <= free // 3
    return _TRAIN_DENSE_CACHE


def dequant_weight(weight: FP4Params, dtype=torch.float16) -> torch.Tensor:
    """Dense [out, in] view of a quantized parameter on its own device (reference op ``xe_linear.dequant``)."""
    n, k = weight._shape
    if weight.data.device.type == "cuda":
        from ..ops import hip
        if weight.qlayout == "t16":
            if _TRAIN_DENSE_CACHE:
                # opt-in (BIGDL_TRAIN_DENSE_CACHE=1): keep the dense copy next to the low-bit one.  A QLoRA step
                # dequantizes every frozen weight twice (forward and backward: 9 % of the step's kernel time on
                # Llama-2-7B); 2 bytes per weight is affordable on a 288 GB device when speed matters more than footprint.
                d = weight.data
                key = (dtype, d.data_ptr(), d._version)
                hit = getattr(weight, "_dense_cache", None)
                if hit is None or hit[0] != key:
                    hit = weight._dense_cache = (key, hip.dequant_t16(d, weight.qtype, n, k, dtype=dtype))
                return hit[1]
            return hip.dequant_t16(weight.data, weight.qtype, n, k, dtype=dtype)
        raw = weight.data.view(torch.uint8) if weight.data.dtype != torch.uint8 else weight.data
        if weight.qtype in (Q.IQ2_XXS, Q.IQ2_XS, Q.IQ1_S, Q.IQ1_M):
            return host.dequantize(raw.cpu(), weight.qtype, (n, k)).to(weight.data.device, dtype)
        return hip.dequant_rows(raw, weight.qtype, n, k, dtype=dtype)
    raw = weight.data.view(torch.uint8) if weight.data.dtype != torch.uint8 else weight.data
    return host.dequantize(raw, weight.qtype, (n, k)).to(dtype)


def dequant_weight_rows(weight: FP4Params, row0: int, nrows: int, dtype=torch.float16) -> torch.Tensor:
    """Dense rows [row0, row0 + nrows) of a quantized parameter (one expert of a row-stacked expert matrix)."""
    n, k = weight._shape
    raw = weight.data.view(torch.uint8) if weight.data.dtype != torch.uint8 else weight.data
    if weight.data.device.type == "cuda" and weight.qlayout == "t16" and row0 % 16 == 0:
        from ..ops import hip
        return hip.dequant_t16(weight.data, weight.qtype, nrows, k, dtype=dtype, total_rows=n, row0=row0)
    if weight.data.device.type == "cuda" and weight.qlayout == "t16":
        return dequant_weight(weight, dtype)[row0:row0 + nrows]
    row_bytes = raw.numel() // n
    sl = raw.view(-1)[row0 * row_bytes:(row0 + nrows) * row_bytes]
    if sl.device.type == "cuda":
        from ..ops import hip
        if weight.qtype in (Q.IQ2_XXS, Q.IQ2_XS, Q.IQ1_S, Q.IQ1_M):
            return host.dequantize(sl.cpu(), weight.qtype, (nrows, k)).to(sl.device, dtype)
        return hip.dequant_rows(sl, weight.qtype, nrows, k, dtype=dtype)
    return host.dequantize(sl, weight.qtype, (nrows, k)).to(dtype)


def lowbit_rows(xh, w, qtype, n, k, bias=None, out_dtype=None):
    """More than GEMM_MAX_ROWS fp16 rows times a T16 weight on this repository's own kernels: the prompt-sized fused GEMM
    where it applies (K a multiple of 128), otherwise the mid-size fused GEMM GEMM_MAX_ROWS rows at a time."""
    from ..ops import hip
    if hip.gemm_prompt_supported(qtype, k):
        return hip.gemm_prompt(xh, w, qtype, n, k, bias=bias, out_dtype=out_dtype)
    m = xh.shape[0]
    y = torch.empty(m, n, dtype=out_dtype or torch.float16, device=xh.device)
    for r0 in range(0, m, GEMM_MAX_ROWS):
        hip.gemm_lowbit(xh[r0:r0 + GEMM_MAX_ROWS], w, qtype, n, k, bias=bias, out=y[r0:r0 + GEMM_MAX_ROWS])
    return y


def lowbit_matmul(x_2d: torch.Tensor, weight: FP4Params, out_features: int, for_training: bool = False) -> torch.Tensor:
    """``x_2d @ dequant(weight)^T`` -> dtype of x_2d (reference op ``xe_linear.forward_new``).  ``for_training``: the
    forward of an autograd node whose backward will expand the weight anyway (bf16 activations): prompt-sized row
    counts then take the dense product in the activations' own dtype instead of the fp16 fused kernel."""
    n, k = weight._shape
    if weight.data.device.type == "cuda":
        from ..ops import hip
        m = x_2d.shape[0]
        if weight.qlayout == "t16" and m <= GEMM_MAX_ROWS:
            xh = x_2d if x_2d.dtype == torch.float16 else x_2d.to(torch.float16)
            if xh.stride(1) != 1 or xh.stride(0) % 8 or xh.data_ptr() % 16:
                xh = xh.contiguous()
            if m <= GEMV_MAX_ROWS:
                return hip.gemv_lowbit(xh, weight.data, weight.qtype, n, k, out_dtype=x_2d.dtype)
            return hip.gemm_lowbit(xh, weight.data, weight.qtype, n, k, out_dtype=x_2d.dtype)
        if weight.qlayout == "t16" and (not for_training or _TRAIN_FUSED_FORWARD) and getattr(weight, "_dense_cache", None) is None:
            # prompt-sized rows: the fused kernels take every row count (reference: one native op,
            # low_bit_linear.py:657-668); no fp16 copy of the weight is written and no library GEMM runs.  (Training keeps
            # the dense product below: its backward needs the expanded weight anyway, and its activations are bf16 --
            # measured 0.106 against 0.154 s per QLoRA step on Llama-2-7B.  Inside an autograd node's forward grad mode is
            # off, so the caller says so explicitly.)
            if x_2d.dtype == torch.float16:
                xh = x_2d
            elif x_2d.dtype == torch.bfloat16:
                xh = x_2d.clamp(-65504.0, 65504.0).to(torch.float16)      # (bf16 reaches further than fp16)
            else:
                xh = x_2d.to(torch.float16)
            if xh.stride(1) != 1 or xh.stride(0) % 8 or xh.data_ptr() % 16:
                xh = xh.contiguous()
            return lowbit_rows(xh, weight.data, weight.qtype, n, k, out_dtype=x_2d.dtype)
        cdt = x_2d.dtype if x_2d.dtype in (torch.float16, torch.bfloat16) else torch.float16
        w = dequant_weight(weight, cdt)
        xc = x_2d.to(cdt)
        if m > 4096:
            # 4096 rows at a time: hipBLASLt is 1.5 - 2.4x slower per row right above 4096 rows than at 4096
            y = torch.empty(m, n, dtype=cdt, device=xc.device)
            for r in range(0, m, 4096):
                torch.mm(xc[r:r + 4096], w.t(), out=y[r:r + 4096])
            return y.to(x_2d.dtype)
        return F.linear(xc, w).to(x_2d.dtype)
    # CPU: quantized matmul for few rows, dequantize + BLAS above the threshold
    raw = weight.data.view(torch.uint8) if weight.data.dtype != torch.uint8 else weight.data
    if x_2d.shape[0] >= TORCH_LINEAR_THRESHOLD:
        return F.linear(x_2d.float(), host.dequantize(raw, weight.qtype, (n, k))).to(x_2d.dtype)
    return host.matmul_q(x_2d, raw, weight.qtype, n, k).to(x_2d.dtype)


class MatMulLowBit(torch.autograd.Function):
    """Frozen-weight low-bit matmul with activation gradient: the QLoRA building block.
    backward = grad_out @ dequant(W) (weights get no gradient)."""

    @staticmethod
    def forward(ctx, A, weight, output_size):
        ctx.weight = weight if A.requires_grad else None
        return lowbit_matmul(A, weight, output_size, for_training=A.requires_grad)

    @staticmethod
    def backward(ctx, grad_output):
        grad_A = None
        if ctx.needs_input_grad[0] and ctx.weight is not None:
            w = dequant_weight(ctx.weight, grad_output.dtype if grad_output.dtype != torch.float32 else torch.float32)
            grad_A = torch.matmul(grad_output, w.to(grad_output.dtype))
        return grad_A, None, None


MatMulLowBitCPU = MatMulLowBit


def reshape_lm_head_input(x):
    """Prefill only needs the logits of the last position (reference :262-269)."""
    if x.dim() > 3:
        x = x.reshape([-1, x.shape[-2], x.shape[-1]])
    shape = list(x.size())
    if shape[1] > 10:
        shape[1] = 1
        x = x[:, -1, :].view(shape)
    return x


def _all_reduce(t, group):
    import torch.distributed as dist
    if hasattr(group, "all_reduce"):      # vLLM style group coordinator
        return group.all_reduce(t)
    dist.all_reduce(t, group=group)
    return t


class LowBitLinear(nn.Linear):
    def __init__(self, input_features, output_features, qtype, bias=True, conver_to_half=True, mp_group=None,
                 optimize_lm_head=False, act_order=False, enable_scale_search=False):
        # nn.Linear.__init__ would draw a full fp32 kaiming-uniform weight that the converter replaces right away
        # (6.5e9 random numbers for a 7B model: most of the conversion / load time): allocate without initialising
        nn.Module.__init__(self)
        self.in_features, self.out_features = input_features, output_features
        self.weight = FP4Params(torch.empty(output_features, input_features), requires_grad=False, quantized=False,
                                _shape=None, qtype=qtype, enable_scale_search=enable_scale_search)
        if bias:
            self.bias = nn.Parameter(torch.zeros(output_features))
        else:
            self.register_parameter("bias", None)
        self.in_len = input_features
        self.out_len = output_features
        self.weight_shape = (self.out_len, self.in_len)
        self.weight_length = self.out_len * self.in_len
        self.qtype = qtype
        self.conver_to_half = conver_to_half
        self.mp_group = mp_group
        self.compute_dtype = None  # only for training
        self.optimize_lm_head = optimize_lm_head
        self.act_order = act_order
        if act_order:
            self.register_buffer("g_idx_map", torch.arange(self.in_len, dtype=torch.int64))

    # the format of the weight as it is stored: with the mixed formats (mixed_fp4 / mixed_fp8) every tensor settles on
    # one of two formats when it is quantized, and kernels have to be told that one, not the name of the mixture
    @property
    def qtype(self):
        w = self._parameters.get("weight")
        if isinstance(w, FP4Params) and w.quantized:
            return w.qtype
        return self._qtype

    @qtype.setter
    def qtype(self, value):
        self._qtype = value

    def forward(self, x: torch.Tensor):
        is_training = self.training and not torch.is_inference_mode_enabled()
        if is_training:
            if self.compute_dtype is not None and x.device.type == "cuda":
                x = x.to(self.compute_dtype)
            elif torch.is_autocast_enabled():
                x = x.to(torch.get_autocast_dtype(x.device.type))
        if self.bias is not None and self.bias.dtype != x.dtype:
            self.bias.data = self.bias.data.to(x.dtype)
        if self.optimize_lm_head and not (self.training and torch.is_grad_enabled()):
            x = reshape_lm_head_input(x)       # inference only: a training step needs the logits of every position
        new_shape = x.shape[:-1] + (self.out_len,)
        if 0 in x.shape:
            return torch.empty(new_shape, dtype=x.dtype, device=x.device)
        if self.act_order:
            x = x[..., self.g_idx_map]
        x_2d = x.contiguous().view(-1, x.shape[-1])
        if not self.weight.quantized:
            self.weight = self.weight.to(self.weight.data.device)  # lazily quantize (CPU only)
        if is_training and x_2d.requires_grad:
            result = MatMulLowBit.apply(x_2d, self.weight, self.out_len)
        else:
            result = lowbit_matmul(x_2d, self.weight, self.out_len)
        result = result.view(new_shape)
        if self.mp_group is not None:
            result = _all_reduce(result, self.mp_group)
        if self.bias is not None:
            result = result + self.bias
        return result.to(x.dtype)

    def extra_repr(self):
        return f"in={self.in_len}, out={self.out_len}, qtype={Q.qtype_name(self.qtype)}, bias={self.bias is not None}"


class _HalfLinear(nn.Linear):
    _dtype = torch.float16

    def __init__(self, input_features, output_features, bias=True, mp_group=None, weight_type=1,
                 optimize_lm_head=False):
        nn.Module.__init__(self)          # (no random init of a weight the converter replaces)
        self.in_features, self.out_features = input_features, output_features
        self.weight = nn.Parameter(torch.empty(output_features, input_features, dtype=self._dtype), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(output_features, dtype=self._dtype), requires_grad=False)
        else:
            self.register_parameter("bias", None)
        self.in_len = input_features
        self.out_len = output_features
        self.weight_shape = (self.out_len, self.in_len)
        self.weight_length = self.out_len * self.in_len
        self.qtype = Q.FP16 if self._dtype == torch.float16 else Q.BF16
        self.mp_group = mp_group
        self.weight_type = weight_type
        self.optimize_lm_head = optimize_lm_head
        self.compute_dtype = None

    def forward(self, x: torch.Tensor):
        if self.optimize_lm_head and not (self.training and torch.is_grad_enabled()):
            x = reshape_lm_head_input(x)
        if self.weight.dtype != self._dtype:
            self.weight.data = self.weight.data.to(self._dtype)
        if self.bias is not None and self.bias.dtype != self._dtype:
            self.bias.data = self.bias.data.to(self._dtype)
        result = self._skinny(x)
        if result is None:
            result = F.linear(x.to(self._dtype), self.weight)
            if self.mp_group is None and self.bias is not None:
                result = result + self.bias
        if self.mp_group is not None:
            result = _all_reduce(result, self.mp_group)
            if self.bias is not None:
                result = result + self.bias
        return result.to(x.dtype)

    def _skinny(self, x):
        """The decode-sized product (<= 16 rows) on this repo's streaming fp16 kernel (csrc/hip/gemv_f16.hip; bias in
        its epilogue unless a tensor-parallel sum comes first) instead of a library GEMM.  None: not applicable."""
        if self._dtype != torch.float16 or x.device.type != "cuda" or (torch.is_grad_enabled() and x.requires_grad):
            return None
        rows = x.numel() // x.shape[-1] if x.shape[-1] else 0
        from ..ops import hip
        if not hip.gemv_f16_ok(rows, self.out_len, self.in_len) or not self.weight.is_contiguous():
            return None
        x2 = x.reshape(rows, self.in_len).to(torch.float16)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        y = hip.gemv_f16(x2, self.weight.data, bias=self.bias.data if self.bias is not None and self.mp_group is None else None)
        return y.view(*x.shape[:-1], self.out_len)


class FP16Linear(_HalfLinear):
    _dtype = torch.float16


class BF16Linear(_HalfLinear):
    _dtype = torch.bfloat16


class vLLMLowBitLinear(LowBitLinear):
    def forward(self, x: torch.Tensor):
        return super().forward(x.to(torch.bfloat16) if self.qtype == Q.NF4 and x.dtype == torch.float32 else x), None


class vLLMFP16Linear(FP16Linear):
    def forward(self, x: torch.Tensor):
        return super().forward(x), None


class vLLMBF16Linear(BF16Linear):
    def forward(self, x: torch.Tensor):
        return super().forward(x), None


def tensor_qtype_overrides(model) -> dict:
    """{module name: format name} of the low-bit linears whose stored format is not the one the model was converted
    with (the per-tensor outcome of a mixed format): written next to a low-bit checkpoint, because the packed bytes of
    sym_int4 / fp4 (and sym_int8 / fp8_e4m3) tensors are the same size and nothing else tells them apart."""
    out = {}
    for name, m in model.named_modules():
        if isinstance(m, LowBitLinear) and isinstance(m.weight, FP4Params) and m.weight.quantized and m.weight.qtype != m._qtype:
            out[name] = Q.qtype_name(m.weight.qtype)
    return out


def apply_tensor_qtype_overrides(model, overrides: dict):
    mods = dict(model.named_modules())
    for name, fmt in (overrides or {}).items():
        m = mods.get(name)
        if isinstance(m, LowBitLinear) and isinstance(m.weight, FP4Params):
            m.weight.qtype = Q.qtype_id(fmt)
