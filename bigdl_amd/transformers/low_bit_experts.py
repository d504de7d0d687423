"""Low-bit mixture-of-experts weights for the module (HuggingFace forward) path.

HuggingFace keeps the experts of Mixtral / Qwen2-MoE / ... as fused 3-D parameters (``gate_up_proj`` [E, 2I, H],
``down_proj`` [E, H, I]) that ``nn.Linear`` replacement never sees.  ``LowBitExperts`` stores them as three row-stacked
quantized matrices (gate [E*I, H], up [E*I, H], down [E*H, I]): each is an ordinary ``LowBitLinear`` weight, so
quantization, ``.to(device)`` re-tiling and ``save_low_bit`` / ``load_low_bit`` work unchanged, and on the device expert
``e`` is tile rows ``e*N/16 ...`` of one buffer, which the routed GEMV indexes from the router's output without a
host round trip.  Reference behaviour: per-expert ``LowBitLinear`` modules and the python loops over them in
python/llm/src/ipex_llm/transformers/models/qwen2_moe.py:376-452 (``qwen2moe_moeblock_forward``: top-k loop at one
token, host-side ``xe_linear.get_moe_indexes`` bucketing below 256 tokens, one-hot mask loop above) and
models/phixtral.py:44-104; Mixtral itself runs through the stock HuggingFace expert loop there.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from .. import qtypes as Q
from .low_bit_linear import FP4Params, LowBitLinear, dequant_weight_rows

ROUTED_GEMV_MAX_SLOTS = 64
# inference with many routed rows: grouped dequant-fused MFMA GEMM; BIGDL_MOE_GROUPED_GEMM=0 falls back to one dense
# GEMM pair per expert (kept as an independent reference path)
GROUPED_GEMM = os.environ.get("BIGDL_MOE_GROUPED_GEMM", "1") != "0"


def is_fused_experts(module) -> bool:
    g, d = getattr(module, "gate_up_proj", None), getattr(module, "down_proj", None)
    if not (isinstance(g, nn.Parameter) and isinstance(d, nn.Parameter) and g.dim() == 3 and d.dim() == 3):
        return False
    # the [E, 2I, H] / [E, H, I] layout with a plain activation; experts with biases, interleaved gate / up columns or a
    # clamped activation of their own (GPT-OSS) keep their dense parameters
    if not hasattr(module, "act_fn") or getattr(module, "gate_up_proj_bias", None) is not None:
        return False
    # gate_up [E, 2I, H] and down [E, H, I] (Llama-4 keeps the transposes, [E, H, 2I] / [E, I, H], and is called with
    # rows already grouped by expert), and a forward that takes (hidden_states, top_k_index, top_k_weights)
    if g.shape[0] != d.shape[0] or g.shape[2] != d.shape[1] or g.shape[1] != 2 * d.shape[2]:
        return False
    import inspect
    try:
        params = list(inspect.signature(type(module).forward).parameters)
    except (TypeError, ValueError):
        return False
    return len(params) == 4


class LowBitExperts(nn.Module):
    def __init__(self, num_experts, hidden_dim, intermediate_dim, qtype, act_fn, enable_scale_search=False):
        super().__init__()
        self.num_experts, self.hidden_dim, self.intermediate_dim, self.qtype = num_experts, hidden_dim, intermediate_dim, qtype
        self.act_fn = act_fn
        mk = lambda i, o: LowBitLinear(i, o, qtype, bias=False, enable_scale_search=enable_scale_search)
        with torch.device("meta"):
            self.gate_proj = mk(hidden_dim, num_experts * intermediate_dim)
            self.up_proj = mk(hidden_dim, num_experts * intermediate_dim)
            self.down_proj = mk(intermediate_dim, num_experts * hidden_dim)

    @classmethod
    def from_experts(cls, module, qtype, convert_shape_only=False, imatrix=None, enable_scale_search=False):
        e, two_i, h = module.gate_up_proj.shape
        i = two_i // 2
        new = cls(e, h, i, qtype, module.act_fn, enable_scale_search)
        device = module.gate_up_proj.device
        gu = module.gate_up_proj.data
        parts = {"gate_proj": (gu[:, :i, :].reshape(e * i, h), h), "up_proj": (gu[:, i:, :].reshape(e * i, h), h),
                 "down_proj": (module.down_proj.data.reshape(e * h, i), i)}
        for name, (w, k) in parts.items():
            getattr(new, name)._parameters["weight"] = FP4Params(
                data=w.contiguous() if device.type != "meta" else w, requires_grad=False, quantized=False, _shape=None,
                convert_shape_only=convert_shape_only, qtype=qtype, imatrix=imatrix, in_features=k,
                enable_scale_search=enable_scale_search).to(device)
        new.eval()
        new.requires_grad_(False)
        return new

    def _act_name(self):
        name = type(self.act_fn).__name__.lower()
        if "silu" in name or "swish" in name:
            return "silu"
        if "gelu" in name:
            return "gelu_tanh" if getattr(self.act_fn, "approximate", "none") == "tanh" or "tanh" in name or "new" in name \
                else "gelu"
        if "relu" in name:
            return "relu"
        return None

    def forward(self, hidden_states, top_k_index, top_k_weights):
        """hidden_states [T, H]; top_k_index / top_k_weights [T, k] -> [T, H]"""
        t, h = hidden_states.shape
        k = top_k_index.shape[-1]
        e, i = self.num_experts, self.intermediate_dim
        gate, up, down = self.gate_proj.weight, self.up_proj.weight, self.down_proj.weight
        act = self._act_name()
        if hidden_states.device.type == "cuda" and gate.qlayout == "t16" and t * k <= ROUTED_GEMV_MAX_SLOTS \
                and act is not None and i % 16 == 0 and h % 16 == 0 and not torch.is_grad_enabled():
            from ..ops import hip
            x = hidden_states if hidden_states.dtype == torch.float16 else hidden_states.to(torch.float16)
            ids = top_k_index.reshape(-1).to(torch.int32).contiguous()
            a = hip.gemv_lowbit_moe(x.contiguous(), gate.data, self.qtype, e, i, h, ids, w2=up.data, slot_xdiv=k, act=act)
            y = hip.gemv_lowbit_moe(a, down.data, self.qtype, e, h, i, ids,
                                    slot_scale=top_k_weights.reshape(-1).float().contiguous())
            return y.view(t, k, h).sum(1).to(hidden_states.dtype)
        if hidden_states.device.type == "cuda" and gate.qlayout == "t16" and act is not None and i % 16 == 0 \
                and h % 16 == 0 and not torch.is_grad_enabled() and GROUPED_GEMM:
            from ..models.moe import grouped_moe_ffn
            x = hidden_states if hidden_states.dtype == torch.float16 else hidden_states.to(torch.float16)
            return grouped_moe_ffn(x.contiguous(), top_k_index, top_k_weights, gate.data, up.data, down.data, self.qtype,
                                   e, i, h, act).to(hidden_states.dtype)
        # bucket the (token, slot) pairs by expert; one dense GEMM pair per expert that received tokens
        flat = top_k_index.reshape(-1)
        order = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=e).tolist()
        tok = torch.div(order, k, rounding_mode="floor")
        cdt = hidden_states.dtype if hidden_states.device.type == "cuda" and hidden_states.dtype in (
            torch.float16, torch.bfloat16) else (torch.float16 if hidden_states.device.type == "cuda" else torch.float32)
        xs = hidden_states[tok].to(cdt)
        ws = top_k_weights.reshape(-1)[order].to(cdt)
        out = torch.zeros(t, h, dtype=cdt, device=hidden_states.device)
        off = 0
        for ei, c in enumerate(counts):
            if c == 0:
                continue
            xe = xs[off:off + c]
            g = torch.nn.functional.linear(xe, dequant_weight_rows(gate, ei * i, i, cdt))
            u = torch.nn.functional.linear(xe, dequant_weight_rows(up, ei * i, i, cdt))
            y = torch.nn.functional.linear(self.act_fn(g) * u, dequant_weight_rows(down, ei * h, h, cdt))
            out.index_add_(0, tok[off:off + c], y * ws[off:off + c, None])
            off += c
        return out.to(hidden_states.dtype)

    def extra_repr(self):
        return (f"num_experts={self.num_experts}, hidden={self.hidden_dim}, intermediate={self.intermediate_dim}, "
                f"qtype={Q.qtype_name(self.qtype) if self.qtype in Q._id_to_name else self.qtype}")
