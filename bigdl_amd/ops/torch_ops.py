"""``torch.library`` registrations of the native ops (namespace ``bigdl_amd``) with fake (meta) implementations, so
graphs containing them can be traced / compiled / exported.

Reference: python/llm/src/ipex_llm/transformers/xpu_ops.py:23-162 registers fake kernels for the closed
``ipex_llm::*`` ops (forward_new, dequant, mlp_forward_xpu, rwkv kernels, dequantize_rows, rms_norm, layer_norm,
rotary_*_inplaced, sdp*, quantize/dequantize_key_value, ...).  The real implementations here are the HIP kernels in
``bigdl_amd/csrc/hip`` reached through ``bigdl_amd.ops.hip``.
"""
from __future__ import annotations

import torch

from . import hip

_lib_defined = False


def register():
    global _lib_defined
    if _lib_defined:
        return
    _lib_defined = True
    op = torch.library.custom_op

    @op("bigdl_amd::forward_new", mutates_args=())
    def forward_new(x: torch.Tensor, weight: torch.Tensor, qtype: int, out_features: int) -> torch.Tensor:
        x2 = x.reshape(-1, x.shape[-1])
        xh = x2 if x2.dtype == torch.float16 else x2.to(torch.float16)
        if x2.shape[0] <= 16:
            y = hip.gemv_lowbit(xh.contiguous(), weight, qtype, out_features, x.shape[-1], out_dtype=torch.float16)
        elif x2.shape[0] <= 256:
            y = hip.gemm_lowbit(xh.contiguous(), weight, qtype, out_features, x.shape[-1], out_dtype=torch.float16)
        else:
            y = torch.nn.functional.linear(xh, hip.dequant_t16(weight, qtype, out_features, x.shape[-1]))
        return y.to(x.dtype).view(*x.shape[:-1], out_features)

    @forward_new.register_fake
    def _(x, weight, qtype, out_features):
        return torch.empty([*x.shape[:-1], out_features], dtype=x.dtype, device=x.device)

    @op("bigdl_amd::dequant", mutates_args=())
    def dequant(x: torch.Tensor, weight: torch.Tensor, qtype: int, out_features: int) -> torch.Tensor:
        dt = x.dtype if x.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float16
        return hip.dequant_t16(weight, qtype, out_features, x.shape[-1], dtype=dt)

    @dequant.register_fake
    def _(x, weight, qtype, out_features):
        return torch.empty([out_features, x.shape[-1]], dtype=x.dtype, device=x.device)

    @op("bigdl_amd::mlp_forward", mutates_args=())
    def mlp_forward(x: torch.Tensor, weight1: torch.Tensor, weight2: torch.Tensor, batch_size: int, state_size: int,
                    output_size: int, act_type: int, qtype: int) -> torch.Tensor:
        act = {1: "silu", 2: "gelu", 3: "gelu_tanh", 4: "relu"}[act_type]
        x2 = x.reshape(batch_size, state_size)
        xh = x2 if x2.dtype == torch.float16 else x2.to(torch.float16)
        y = hip.gemv_lowbit(xh.contiguous(), weight1, qtype, output_size, state_size, w2=weight2, act=act)
        return y.to(x.dtype)

    @mlp_forward.register_fake
    def _(x, weight1, weight2, batch_size, state_size, output_size, act_type, qtype):
        return torch.empty([batch_size, output_size], dtype=x.dtype, device=x.device)

    @op("bigdl_amd::dequantize_rows", mutates_args=())
    def dequantize_rows(x: torch.Tensor, weight: torch.Tensor, qtype: int, state_size: int, output_size: int) -> torch.Tensor:
        return hip.dequant_rows(weight, qtype, output_size, state_size, ids=x, dtype=torch.float32)

    @dequantize_rows.register_fake
    def _(x, weight, qtype, state_size, output_size):
        return torch.empty([*x.shape, state_size], dtype=torch.float32, device=weight.device)

    @op("bigdl_amd::rms_norm", mutates_args=())
    def rms_norm(weight: torch.Tensor, x: torch.Tensor, eps: float) -> torch.Tensor:
        return hip.rms_norm(x.contiguous(), weight, eps)

    @rms_norm.register_fake
    def _(weight, x, eps):
        return torch.empty_like(x)

    @op("bigdl_amd::layer_norm", mutates_args=())
    def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
        return hip.layer_norm(x, weight, bias, eps)

    @layer_norm.register_fake
    def _(x, weight, bias, eps):
        return torch.empty_like(x)

    @op("bigdl_amd::rotary_half_inplaced", mutates_args=("query", "key"))
    def rotary_half_inplaced(inv_freq: torch.Tensor, position_ids: torch.Tensor, query: torch.Tensor,
                             key: torch.Tensor) -> None:
        hip.rope_inplace(query, position_ids=position_ids, inv_freq=inv_freq, interleaved=False)
        hip.rope_inplace(key, position_ids=position_ids, inv_freq=inv_freq, interleaved=False)

    @op("bigdl_amd::rotary_two_inplaced", mutates_args=("query", "key"))
    def rotary_two_inplaced(inv_freq: torch.Tensor, position_ids: torch.Tensor, query: torch.Tensor,
                            key: torch.Tensor) -> None:
        hip.rope_inplace(query, position_ids=position_ids, inv_freq=inv_freq, interleaved=True)
        hip.rope_inplace(key, position_ids=position_ids, inv_freq=inv_freq, interleaved=True)

    @op("bigdl_amd::rotary_half_with_cache_inplaced", mutates_args=("query", "key"))
    def rotary_half_with_cache_inplaced(query: torch.Tensor, key: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> None:
        hip.rope_inplace(query, cos=cos, sin=sin, rot_dim=cos.shape[-1])
        hip.rope_inplace(key, cos=cos, sin=sin, rot_dim=cos.shape[-1])

    @op("bigdl_amd::mlp_silu_mul_inplaced", mutates_args=("gate",))
    def mlp_silu_mul_inplaced(gate: torch.Tensor, up: torch.Tensor) -> None:
        hip.act_mul(gate, up, "silu", out=gate)

    @op("bigdl_amd::attn_softmax_inplaced", mutates_args=("attn",))
    def attn_softmax_inplaced(attn: torch.Tensor) -> None:
        hip.softmax_inplace(attn)

    @op("bigdl_amd::quantize_key_value", mutates_args=("key_output", "value_output"))
    def quantize_key_value(key: torch.Tensor, value: torch.Tensor, key_output: torch.Tensor,
                           value_output: torch.Tensor) -> None:
        hip.to_e5m2(key, key_output)
        hip.to_e5m2(value, value_output)

    @op("bigdl_amd::dequantize_key_value", mutates_args=("key_output", "value_output"))
    def dequantize_key_value(key: torch.Tensor, value: torch.Tensor, key_output: torch.Tensor,
                             value_output: torch.Tensor) -> None:
        hip.from_e5m2(key, key_output)
        hip.from_e5m2(value, value_output)

    @op("bigdl_amd::repetition_penalty_logits_process_inplaced", mutates_args=("scores",))
    def repetition_penalty(scores: torch.Tensor, input_ids: torch.Tensor, penalty: float) -> None:
        hip.repetition_penalty_inplace(scores, input_ids, penalty)

    @op("bigdl_amd::moe_softmax_topk", mutates_args=())
    def moe_softmax_topk(logits: torch.Tensor, top_k: int, norm_topk_prob: bool) -> tuple[torch.Tensor, torch.Tensor]:
        ids, w = hip.moe_route(logits.contiguous(), top_k, norm_topk_prob)
        return ids, w

    @moe_softmax_topk.register_fake
    def _(logits, top_k, norm_topk_prob):
        t = logits.shape[0]
        return (torch.empty([t, top_k], dtype=torch.int32, device=logits.device),
                torch.empty([t, top_k], dtype=torch.float32, device=logits.device))


def registered_ops():
    register()
    return sorted(n for n in dir(torch.ops.bigdl_amd) if not n.startswith("_"))
