"""Structural rewrites and forward bindings for HuggingFace decoder models, plus the bridge from an HF
model to the native decoder engine.

The reference keeps one file of forwards per architecture (python/llm/src/ipex_llm/transformers/models/*.py,
bound in convert.py:1251-2027).  HF's current decoder layers share one calling convention
(``self_attn(hidden_states, position_embeddings, attention_mask, past_key_values, **kw)``), so the bindings
here are generic:

* ``merge_qkv``        q/k/v projections -> one ``qkv_proj`` (reference models/common.py:22-52)
* ``attention_forward`` merged projection, in-place RoPE kernel, HF cache update, HF attention interface
* ``mlp_forward``      fused gate/up GEMV + activation for few rows (reference models/common.py:118-141)
* ``rms_norm_forward`` RMSNorm kernel (reference models/common.py:156-174, gemma variant gemma.py:61-72)
* ``layer_norm_forward`` LayerNorm kernel (reference models/common.py:177-188)
* ``generate`` of supported families is served by ``NativeDecoder`` (hipGraph decode) when the call is plain
  greedy/sampling generation on the GPU.
"""
from __future__ import annotations

import logging
import os
import sys
import inspect
import types

import torch
import torch.nn.functional as F
from torch import nn

from .. import qtypes as Q

logger = logging.getLogger("bigdl_amd")

# model types whose attention modules follow the llama layout (q_proj/k_proj/v_proj/o_proj + rotary cos/sin)
MERGE_QKV_TYPES = {"llama", "mistral", "qwen2", "qwen3", "gemma", "gemma2", "stablelm", "starcoder2", "minicpm",
                   "mixtral", "qwen2_moe", "olmo", "cohere", "internlm3", "yi", "glm", "granite",
                   "qwen2_vl", "qwen2_vl_text", "qwen2_5_vl", "qwen2_5_vl_text"}
# multimodal wrappers whose language model the native engine can drive: the wrapper's own forward merges image features
# into the token embeddings and computes the (3-axis) rotary positions, the engine takes both from there
VL_ENGINE_TYPES = {"qwen2_vl", "qwen2_5_vl"}
NATIVE_ENGINE_TYPES = {"llama", "mistral", "qwen2", "gemma", "gemma2", "phi3", "mixtral", "qwen2_moe", "yi", "granite"}


PREFILL_HEAD_DIMS = (64, 72, 80, 96, 128, 256)      # ops/hip.py attn_prefill

def _is_cuda(t):
    return t.device.type == "cuda"


# ------------------------------------------------------------------------------------------------
# structural rewrites
# ------------------------------------------------------------------------------------------------
def merge_linear(linears):
    """Concatenate nn.Linear layers along the output dim."""
    new_weight = torch.cat([l.weight.data for l in linears], dim=0)
    bias = None
    if linears[0].bias is not None:
        bias = torch.cat([l.bias.data for l in linears], dim=0)
    new = nn.Linear(0, 0, bias=False)
    new.weight = nn.Parameter(new_weight, requires_grad=False)
    new.in_features, new.out_features = new_weight.size(1), new_weight.size(0)
    if bias is not None:
        new.bias = nn.Parameter(bias, requires_grad=False)
    return new


def merge_qkv_base(module: nn.Module, attention_class):
    name = attention_class if isinstance(attention_class, str) else attention_class.__name__
    if module.__class__.__name__ == name and all(isinstance(getattr(module, p, None), nn.Linear)
                                                   for p in ("q_proj", "k_proj", "v_proj")):
        module.qkv_proj = merge_linear([module.q_proj, module.k_proj, module.v_proj])
        module._merged_sizes = (module.q_proj.out_features, module.k_proj.out_features, module.v_proj.out_features)
        del module.q_proj, module.k_proj, module.v_proj


def _attention_classes(model):
    out = set()
    for m in model.modules():
        if all(hasattr(m, p) for p in ("q_proj", "k_proj", "v_proj", "o_proj")) and hasattr(m, "layer_idx"):
            out.add(m.__class__)
    return out


def pad_mlp_intermediate(module, multiple: int = 128):
    """Zero-pad the intermediate dimension of a gated MLP (gate_proj / up_proj rows, down_proj columns) to a multiple
    of ``multiple`` so that its three linears qualify for the tiled low-bit formats.  Exact: a padded channel computes
    act(0) * 0 = 0 and meets a zero column of down_proj.  (Reference: models/qwen2.py:199-230 ``padding_mlp`` for
    Qwen1.5-14B's 13696 channels.)"""
    g, u, d = (getattr(module, n, None) for n in ("gate_proj", "up_proj", "down_proj"))
    if not all(type(x) is nn.Linear for x in (g, u, d)):
        return
    inter = g.out_features
    if inter % multiple == 0 or u.out_features != inter or d.in_features != inter:
        return
    new = (inter + multiple - 1) // multiple * multiple

    def grown(t, shape, fill):
        out = torch.zeros(shape, dtype=t.dtype, device=t.device)       # (meta skeletons of load_low_bit: shapes only)
        if t.device.type != "meta":
            fill(out, t)
        return nn.Parameter(out, requires_grad=False)
    for lin in (g, u):
        lin.weight = grown(lin.weight.data, (new, lin.in_features), lambda o, t: o[:inter].copy_(t))
        if lin.bias is not None:
            lin.bias = grown(lin.bias.data, (new,), lambda o, t: o[:inter].copy_(t))
        lin.out_features = new
    d.weight = grown(d.weight.data, (d.out_features, new), lambda o, t: o[:, :inter].copy_(t))
    d.in_features = new
    module._bigdl_padded_from = inter


def fold_norm_head(model):
    """Baichuan2's ``NormHead`` L2-normalises the rows of the output embedding on the fly; at inference the weights are
    fixed, so the normalisation is applied once and the module becomes a plain linear that the low-bit converter
    can take (reference convert.py:908-925)."""
    head = getattr(model, "lm_head", None)
    if head is None or head.__class__.__name__ != "NormHead" or not hasattr(head, "weight"):
        return
    w = head.weight.data
    lin = nn.Linear(w.shape[1], w.shape[0], bias=False, device=w.device, dtype=w.dtype)
    if w.device.type != "meta":
        lin.weight.data = F.normalize(w.float()).to(w.dtype)
    model.lm_head = lin


def optimize_pre(model, qtype=None):
    cfg = getattr(model, "config", None)
    mt = getattr(cfg, "model_type", None)
    from .remote_code import pre_process_remote_code
    pre_process_remote_code(model)            # e.g. partial-LoRA linears of InternLM-XComposer2 become plain linears
    fold_norm_head(model)
    if qtype is None or qtype in Q.T16_QTYPES:
        model.apply(pad_mlp_intermediate)
        # the padded width is what every later reader of the configuration has to see: the native engine sizes its MLP
        # buffers from it, and a saved low-bit checkpoint must describe the tensors it holds (its skeleton is rebuilt
        # from config.json).  The original width is kept beside it.
        padded = {(m._bigdl_padded_from, m.gate_proj.out_features) for m in model.modules()
                  if getattr(m, "_bigdl_padded_from", None) is not None}
        for c in {id(x): x for x in (cfg, getattr(cfg, "text_config", None)) if x is not None}.values():
            for old, new in padded:
                if getattr(c, "intermediate_size", None) == old:
                    c.intermediate_size = new
                    c.bigdl_unpadded_intermediate_size = old
    if mt in MERGE_QKV_TYPES:
        for cls in _attention_classes(model):
            model.apply(lambda m, c=cls: merge_qkv_base(m, c))
    if mt == "rwkv" and hasattr(model, "rwkv") and hasattr(model.rwkv, "_rescale_layers") and not model.training:
        # HF divides output / value weights of deeper blocks in place at the first inference call; that has to
        # happen on the dense weights, before they become quantized byte buffers (reference convert.py:1035-1048)
        model.rwkv._rescale_layers()
    # multimodal wrappers: recurse into the language model with its own config (``model.language_model``, or
    # ``model.model.language_model`` where the wrapper keeps towers and decoder under one base module)
    # (MiniCPM-o also carries a llama-style decoder inside its speech head: ``model.tts.model``, reference
    # convert.py:1030-1041)
    for holder in (model, getattr(model, "model", None), getattr(model, "tts", None)):
        for attr in ("language_model", "llm") if holder is not getattr(model, "tts", None) or holder is None else ("model",):
            sub = getattr(holder, attr, None) if holder is not None else None
            if isinstance(sub, nn.Module) and getattr(getattr(sub, "config", None), "model_type", None) in MERGE_QKV_TYPES \
                    and sub is not model and mt not in MERGE_QKV_TYPES:
                optimize_pre(sub, qtype)
    return model


# ------------------------------------------------------------------------------------------------
# forwards
# ------------------------------------------------------------------------------------------------
def rms_norm_forward(self, hidden_states):
    eps = getattr(self, "variance_epsilon", None)
    if eps is None:
        eps = getattr(self, "eps", 1e-6)
    offset = 1.0 if self.__class__.__name__.startswith("Gemma") else 0.0
    if _is_cuda(hidden_states) and hidden_states.dtype in (torch.float16, torch.bfloat16, torch.float32) \
            and hidden_states.shape[-1] % 8 == 0 and not torch.is_grad_enabled():
        from ..ops import hip
        return hip.rms_norm(hidden_states.contiguous(), self.weight, eps, offset=offset)
    if _is_cuda(hidden_states) and hidden_states.shape[-1] % 8 == 0 and hidden_states.shape[-1] <= 16384 \
            and os.environ.get("BIGDL_FUSED_TRAIN_OPS", "1") != "0":
        from ..transformers.fused_train_ops import RMSNormFn, usable
        if usable(hidden_states) and (hidden_states.requires_grad or self.weight.requires_grad):
            return RMSNormFn.apply(hidden_states.contiguous(), self.weight, float(eps), offset)
    x = hidden_states.float()
    x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return (x * (self.weight.float() + offset)).to(hidden_states.dtype)


def layer_norm_forward(self, hidden_states):
    if _is_cuda(hidden_states) and hidden_states.shape[-1] % 8 == 0 and not torch.is_grad_enabled() \
            and self.weight is not None and hidden_states.dtype in (torch.float16, torch.bfloat16, torch.float32):
        from ..ops import hip
        return hip.layer_norm(hidden_states, self.weight, self.bias, self.eps)
    w, b = self.weight, self.bias
    if w is not None and w.dtype != hidden_states.dtype:       # half activations through a norm kept in fp32 (CPU)
        w, b = w.to(hidden_states.dtype), None if b is None else b.to(hidden_states.dtype)
    return F.layer_norm(hidden_states, self.normalized_shape, w, b, self.eps)


def _act_name(mlp):
    fn = getattr(mlp, "act_fn", None)
    name = type(fn).__name__.lower() if fn is not None else ""
    cfg_act = getattr(getattr(mlp, "config", None), "hidden_act", None) or getattr(getattr(mlp, "config", None), "hidden_activation", None)
    if cfg_act in ("silu", "swish", "gelu", "gelu_pytorch_tanh", "gelu_new", "relu", "gelu_tanh", "gelu_fast"):
        return cfg_act
    if "silu" in name:
        return "silu"
    if "tanh" in name or "newgelu" in name:
        return "gelu_pytorch_tanh"
    if "gelu" in name:
        return "gelu"
    return None


def mlp_forward(self, x):
    """gate/up/down MLP: one fused kernel for gate+up+act when few rows hit T16 weights on the GPU."""
    from ..transformers.low_bit_linear import LowBitLinear
    g, u, d = self.gate_proj, self.up_proj, self.down_proj
    act = _act_name(self)
    rows = x.numel() // x.shape[-1]
    if (_is_cuda(x) and act is not None and rows <= 32 and isinstance(g, LowBitLinear) and isinstance(u, LowBitLinear)
            and g.weight.qlayout == "t16" and u.weight.qlayout == "t16" and g.qtype == u.qtype
            and g.mp_group is None and not torch.is_grad_enabled()):
        from ..ops import hip
        x2 = x.reshape(rows, x.shape[-1])
        xh = x2 if x2.dtype == torch.float16 else x2.to(torch.float16)
        h = hip.gemv_lowbit(xh.contiguous(), g.weight.data, g.qtype, g.out_len, g.in_len, w2=u.weight.data,
                            bias=None if g.bias is None else g.bias.to(torch.float16),
                            bias2=None if u.bias is None else u.bias.to(torch.float16), act=act)
        return d(h.to(x.dtype).view(*x.shape[:-1], g.out_len))
    if _is_cuda(x) and act in ("silu", "swish") and torch.is_grad_enabled() and os.environ.get("BIGDL_FUSED_TRAIN_OPS", "1") != "0":
        from ..transformers.fused_train_ops import SiluMulFn, usable
        gate, up = g(x), u(x)
        if usable(gate, up) and gate.dtype == up.dtype and gate.shape[-1] % 8 == 0 and (gate.requires_grad or up.requires_grad):
            return d(SiluMulFn.apply(gate.contiguous(), up.contiguous()))
        return d(self.act_fn(gate) * up)
    return d(self.act_fn(g(x)) * u(x))


def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary(q, k, cos, sin):
    """RoPE on [B,H,S,D] q/k (views allowed) with HF cos/sin [B,S,rot]; in place on the GPU."""
    rot = cos.shape[-1]
    if _is_cuda(q) and not torch.is_grad_enabled() and q.stride(-1) == 1 and k.stride(-1) == 1:
        from ..ops import hip
        hip.rope_inplace(q, cos=cos, sin=sin, rot_dim=rot)
        hip.rope_inplace(k, cos=cos, sin=sin, rot_dim=rot)
        return q, k
    if _is_cuda(q) and torch.is_grad_enabled() and (q.requires_grad or k.requires_grad) and cos.dim() == 3 \
            and os.environ.get("BIGDL_FUSED_TRAIN_OPS", "1") != "0":
        from ..transformers.fused_train_ops import RopeFn, usable
        if usable(q, k) and cos.dtype == q.dtype:
            return RopeFn.apply(q, cos, sin), RopeFn.apply(k, cos, sin)
    cos_, sin_ = cos.unsqueeze(1), sin.unsqueeze(1)
    if rot == q.shape[-1]:
        return q * cos_ + _rotate_half(q) * sin_, k * cos_ + _rotate_half(k) * sin_
    qr, qp = q[..., :rot], q[..., rot:]
    kr, kp = k[..., :rot], k[..., rot:]
    return (torch.cat([qr * cos_ + _rotate_half(qr) * sin_, qp], -1),
            torch.cat([kr * cos_ + _rotate_half(kr) * sin_, kp], -1))


def attention_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None,
                      **kwargs):
    """Generic llama-layout attention over the merged ``qkv_proj``."""
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    input_shape = hidden_states.shape[:-1]
    hd = self.head_dim
    nq, nk, nv = (s // hd for s in self._merged_sizes)
    qkv = self.qkv_proj(hidden_states).view(*input_shape, nq + nk + nv, hd).transpose(1, 2)
    q, k, v = qkv.split([nq, nk, nv], dim=1)
    if getattr(self, "q_norm", None) is not None:      # qwen3 / gemma3 / olmo2 style per-head norms
        q, k = self.q_norm(q), self.k_norm(k)
    if position_embeddings is not None:
        cos, sin = position_embeddings
        if cos.dim() == 4:                             # [3, B, S, D]: temporal / height / width positions (Qwen2-VL)
            cos, sin = merge_mrope(cos, sin, _mrope_section(self))
        q, k = apply_rotary(q, k, cos, sin)
    if past_key_values is not None:
        if getattr(past_key_values, "wants_query", False):     # SnapKV selection scores the prompt with its last queries
            k, v = past_key_values.update(k, v, self.layer_idx, query_states=q, num_key_value_groups=nq // nk)
        else:
            k, v = past_key_values.update(k, v, self.layer_idx)
    if k.dtype == torch.uint8:
        # byte views of an Fp8RawLayer (K, V^T as e5m2): read in place by the flash kernel, else expanded once
        raw = _fp8_raw_attention(self, q, k, v, attention_mask, 0.0, self.scaling, None,
                                 dict(kwargs, softcap=getattr(self, "attn_logit_softcapping", None)))
        if raw is not None:
            return self.o_proj(raw.reshape(*input_shape, -1)), None
        from ..transformers.kv import fp8_raw_to_dense
        k, v = fp8_raw_to_dense(k, v, q.dtype if q.dtype in (torch.float16, torch.bfloat16) else torch.float16)
    native = _native_attention(self, q, k, v, attention_mask, kwargs)
    if native is not None:
        return self.o_proj(native), None
    mod = sys.modules[self.__class__.__module__]
    eager = getattr(mod, "eager_attention_forward")
    interface = ALL_ATTENTION_FUNCTIONS.get_interface(self.config._attn_implementation, eager)
    extra = {}
    if hasattr(self, "sliding_window"):
        extra["sliding_window"] = self.sliding_window
    if getattr(self, "attn_logit_softcapping", None) is not None:
        extra["softcap"] = self.attn_logit_softcapping
    attn_output, attn_weights = interface(self, q, k, v, attention_mask,
                                          dropout=0.0 if not self.training else getattr(self, "attention_dropout", 0.0),
                                          scaling=self.scaling, **extra, **kwargs)
    attn_output = attn_output.reshape(*input_shape, -1).contiguous()
    return self.o_proj(attn_output), attn_weights


def _mrope_section(module):
    rp = getattr(module.config, "rope_parameters", None) or getattr(module.config, "rope_scaling", None) or {}
    sec = rp.get("mrope_section") or getattr(module, "rope_scaling", {}).get("mrope_section")
    if not sec:
        raise ValueError("3-axis rotary positions without an mrope_section in the model config")
    return list(sec)


def merge_mrope(cos, sin, section):
    """cos / sin [3, B, S, D] (one rotary table per position axis) -> [B, S, D]: frequency block j of each half of the
    head takes its angles from axis j % 3 (reference models/qwen2_vl.py:268-319, HF ``apply_multimodal_rotary_pos_emb``)."""
    sec = list(section) * 2
    pick = lambda t: torch.cat([m[i % 3] for i, m in enumerate(t.split(sec, dim=-1))], dim=-1)   # noqa: E731
    return pick(cos), pick(sin)


def _native_attention(module, q, k, v, attention_mask, kwargs):
    """The MFMA flash-attention kernel for the module path: fp16 tensors on the GPU, head_dim 64 / 72 / 80 / 96 / 128 / 256,
    inference (72 and 80 -- SigLIP-style vision towers, phi-2 -- run in 96-wide tiles without a padded copy).
    HF hands over either no mask (pure causal) or an additive / boolean [B, 1, Sq, Lk] mask that already contains the
    causal and sliding-window structure.  Returns [B, Sq, Hq*D] or None when the case is not covered."""
    if not _is_cuda(q) or q.dtype != torch.float16 or torch.is_grad_enabled() or q.shape[-1] not in PREFILL_HEAD_DIMS \
            or kwargs.get("output_attentions") or module.training:
        return None
    if q.stride(-1) != 1 or k.stride(-1) != 1 or v.stride(-1) != 1 or any(t.stride(i) % 8 for t in (q, k, v) for i in (0, 1, 2)):
        return None
    mask = attention_mask
    if mask is not None:
        if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[-2] != q.shape[2] or mask.shape[-1] != k.shape[2]:
            return None
        if mask.dtype == torch.bool:
            mask = torch.zeros(mask.shape, dtype=torch.float16, device=mask.device).masked_fill_(~mask, float("-inf"))
        elif mask.dtype != torch.float16:
            mask = mask.to(torch.float16)
        if mask.stride(-1) != 1:
            mask = mask.contiguous()
    from ..ops import hip
    softcap = getattr(module, "attn_logit_softcapping", None) or 0.0
    return hip.attn_prefill(q, k, v, scale=module.scaling, causal=mask is None, mask=mask, softcap=softcap)


HIP_ATTENTION = "bigdl_hip"


def _fp8_raw_attention(module, query, key, value, attention_mask, dropout, scaling, is_causal, kwargs):
    from ..ops import hip
    b, hq, sq, d = query.shape
    lk = key.shape[2]
    if not (_is_cuda(query) and query.dtype == torch.float16 and not torch.is_grad_enabled() and d in (64, 96, 128, 256)
            and not module.training and not dropout and kwargs.get("position_bias") is None
            and not kwargs.get("output_attentions") and hq % key.shape[1] == 0 and lk >= sq):
        return None
    mask = attention_mask
    if mask is not None:
        if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[-2] != sq or mask.shape[-1] != lk:
            return None
        if mask.dtype == torch.bool:
            mask = torch.zeros(mask.shape, dtype=torch.float16, device=mask.device).masked_fill_(~mask, float("-inf"))
        elif mask.dtype != torch.float16:
            mask = mask.to(torch.float16)
        if mask.stride(-1) != 1:
            mask = mask.contiguous()
    causal = mask is None and sq > 1 and (is_causal if is_causal is not None else getattr(module, "is_causal", True))
    q = query if (query.stride(-1) == 1 and all(query.stride(i) % 8 == 0 for i in (0, 1, 2))) else query.contiguous()
    softcap = kwargs.get("softcap", getattr(module, "attn_logit_softcapping", None)) or 0.0
    out = hip.attn_prefill(q, key, value, scale=scaling if scaling is not None else d ** -0.5, causal=bool(causal),
                           mask=mask, softcap=softcap, v_transposed=True, kv_e5m2=True)
    return out.view(b, sq, hq, d)


def hip_attention_interface(module, query, key, value, attention_mask, dropout=0.0, scaling=None, is_causal=None, **kwargs):
    """Entry in HuggingFace's attention-function registry (``config._attn_implementation = "bigdl_hip"``): every
    architecture that dispatches through that registry (gpt2, gpt_neox, falcon, phi, starcoder2, opt, whisper, bert,
    ... next to the llama-layout ones) reaches the MFMA flash-attention kernel without a per-family forward -- the
    reference rebinds one attention forward per family instead (python/llm/src/ipex_llm/transformers/models/
    {gpt2,gptneox,phi,starcoder2,whisper,bert}.py).  Cases the kernel does not cover fall through to PyTorch SDPA."""
    from transformers.integrations.sdpa_attention import sdpa_attention_forward
    d = query.shape[-1]
    if key.dtype == torch.uint8:
        # byte views of an Fp8RawLayer: K [B, Hkv, L, D] and V^T [B, Hkv, D, L] as e5m2.  The flash kernel reads them as
        # they lie (prompt chunks and single decode tokens alike); anything it does not cover gets a dense copy.
        out = _fp8_raw_attention(module, query, key, value, attention_mask, dropout, scaling, is_causal, kwargs)
        if out is not None:
            return out, None
        from ..transformers.kv import fp8_raw_to_dense
        key, value = fp8_raw_to_dense(key, value, query.dtype if query.dtype in (torch.float16, torch.bfloat16) else torch.float16)
    ok = (_is_cuda(query) and query.dtype == torch.float16 and not torch.is_grad_enabled() and d in PREFILL_HEAD_DIMS
          and not module.training and not dropout and kwargs.get("position_bias") is None
          and query.dim() == 4 and key.shape[1] > 0 and query.shape[1] % key.shape[1] == 0)
    if ok:
        sq, lk = query.shape[2], key.shape[2]
        causal = attention_mask is None and sq > 1 and (is_causal if is_causal is not None else getattr(module, "is_causal", True))
        if attention_mask is None and sq > 1 and not causal and sq != lk and getattr(module, "is_causal", False):
            ok = False
    if ok:
        class _Shim:                      # what _native_attention reads from the attention module
            pass
        shim = _Shim()
        shim.scaling = scaling if scaling is not None else d ** -0.5
        shim.attn_logit_softcapping = kwargs.get("softcap", getattr(module, "attn_logit_softcapping", None))
        shim.training = False
        q, k, v = query, key, value
        fix = lambda t: t if (t.stride(-1) == 1 and all(t.stride(i) % 8 == 0 for i in (0, 1, 2))) else t.contiguous()  # noqa: E731
        mask = attention_mask
        if mask is None and not causal:
            out = _native_attention_call(fix(q), fix(k), fix(v), shim, None, causal=False)
        else:
            out = _native_attention(shim, fix(q), fix(k), fix(v), mask, {})
        if out is not None:
            return out.view(query.shape[0], query.shape[2], query.shape[1], d), None
    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling,
                                  is_causal=is_causal, **kwargs)


def _native_attention_call(q, k, v, shim, mask, causal):
    from ..ops import hip
    return hip.attn_prefill(q, k, v, scale=shim.scaling, causal=causal, mask=mask, softcap=shim.attn_logit_softcapping or 0.0)


def register_hip_attention(model):
    """Register the kernel in HF's attention registry and select it for ``model`` (and nested configs)."""
    try:
        from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
        from transformers.masking_utils import ALL_MASK_ATTENTION_FUNCTIONS
    except ImportError:
        return False
    if HIP_ATTENTION not in ALL_ATTENTION_FUNCTIONS:
        ALL_ATTENTION_FUNCTIONS.register(HIP_ATTENTION, hip_attention_interface)
        # masks are built as for SDPA: None for pure causal, additive / boolean [B, 1, Sq, Lk] otherwise
        ALL_MASK_ATTENTION_FUNCTIONS.register(HIP_ATTENTION, ALL_MASK_ATTENTION_FUNCTIONS["sdpa"])
    done = False
    if any(getattr(m, "sinks", None) is not None for m in model.modules()):
        # attention sinks (GPT-OSS: one extra logit per head in the softmax): neither the kernel nor SDPA has them, the
        # model keeps the attention implementation it was loaded with
        return False
    for m in model.modules():
        cfg = getattr(m, "config", None)
        # (``_supports_sdpa``: the model class itself declares that an SDPA-style backend computes what its eager
        # attention computes; bigbird_pegasus, for one, goes through the registry but is only right in eager form)
        if cfg is not None and getattr(cfg, "_attn_implementation", None) in ("sdpa", "eager", None) \
                and getattr(type(m), "_supports_sdpa", False) and _dispatches_through_registry(type(m)):
            try:
                cfg._attn_implementation = HIP_ATTENTION
                done = True
            except Exception:
                pass
    return done


_REGISTRY_DISPATCH = {}


def _dispatches_through_registry(cls) -> bool:
    """True when the modeling file of ``cls`` picks its attention through ALL_ATTENTION_FUNCTIONS only; files that
    compare ``_attn_implementation`` with the names of fused implementations (falcon) or never consult the registry
    (bloom-style manual attention) keep their own choice."""
    mod = sys.modules.get(cls.__module__)
    if mod not in _REGISTRY_DISPATCH:
        try:
            import inspect
            src = inspect.getsource(mod)
            import re
            # a comparison with "eager" is harmless (we are one of the non-eager choices); files that branch on the
            # names of specific fused implementations would take their eager path with a mask built for SDPA
            named = re.search(r"_attn_implementation\s*(==|!=|in)\s*[^\n]*(sdpa|flash|flex)", src)
            _REGISTRY_DISPATCH[mod] = "ALL_ATTENTION_FUNCTIONS" in src and named is None
        except Exception:
            _REGISTRY_DISPATCH[mod] = False
    return _REGISTRY_DISPATCH[mod]


# ------------------------------------------------------------------------------------------------
# HF model -> native engine
# ------------------------------------------------------------------------------------------------
def decoder_config_from_hf(cfg):
    from .decoder import DecoderConfig
    mt = cfg.model_type
    hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
    rope_params = getattr(cfg, "rope_parameters", None) or {}
    rope_theta = rope_params.get("rope_theta", getattr(cfg, "rope_theta", 10000.0)) or 10000.0
    rope_scaling = getattr(cfg, "rope_scaling", None)
    if rope_scaling is None and rope_params.get("rope_type", "default") not in ("default", None):
        rope_scaling = dict(rope_params)
    partial = rope_params.get("partial_rotary_factor", getattr(cfg, "partial_rotary_factor", 1.0)) or 1.0
    dc = DecoderConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
        intermediate_size=getattr(cfg, "intermediate_size", 0) or 0, num_layers=cfg.num_hidden_layers,
        num_heads=cfg.num_attention_heads, num_kv_heads=getattr(cfg, "num_key_value_heads", None) or cfg.num_attention_heads,
        head_dim=hd, rms_eps=getattr(cfg, "rms_norm_eps", 1e-6), rope_theta=float(rope_theta),
        rope_scaling=rope_scaling, max_position=getattr(cfg, "max_position_embeddings", 4096),
        tie_word_embeddings=bool(getattr(cfg, "tie_word_embeddings", False)),
        act=getattr(cfg, "hidden_act", None) or getattr(cfg, "hidden_activation", None) or "silu", model_type=mt,
        rot_dim=int(hd * partial) if partial != 1.0 else None)
    if mt in ("gemma", "gemma2"):
        dc.norm_offset = 1.0
        dc.embed_scale = float(cfg.hidden_size) ** 0.5
        dc.act = getattr(cfg, "hidden_activation", None) or "gelu_pytorch_tanh"
    if mt == "gemma2":
        dc.attn_softcap = float(getattr(cfg, "attn_logit_softcapping", 0.0) or 0.0)
        dc.final_softcap = float(getattr(cfg, "final_logit_softcapping", 0.0) or 0.0)
        dc.attn_scale = float(getattr(cfg, "query_pre_attn_scalar", hd)) ** -0.5
        dc.sliding_window = int(getattr(cfg, "sliding_window", 0) or 0)
        dc.post_norms = True
    if mt in ("mistral", "mixtral"):
        dc.sliding_window = int(getattr(cfg, "sliding_window", 0) or 0)
    if mt == "granite":
        # (residual_multiplier has no engine equivalent: native_engine_supported sends such checkpoints to the HF path)
        dc.logit_scale = 1.0 / float(getattr(cfg, "logits_scaling", 1.0))
        dc.embed_scale = float(getattr(cfg, "embedding_multiplier", 1.0))
        dc.attn_scale = float(getattr(cfg, "attention_multiplier", None) or hd ** -0.5)
    if mt == "mixtral":
        dc.num_experts, dc.top_k = cfg.num_local_experts, cfg.num_experts_per_tok
        dc.moe_intermediate_size = cfg.intermediate_size
    if rope_params.get("mrope_section") or (rope_scaling or {}).get("mrope_section"):
        dc.mrope_section = tuple(rope_params.get("mrope_section") or rope_scaling["mrope_section"])
        if (dc.rope_scaling or {}).get("rope_type", (dc.rope_scaling or {}).get("type")) in ("mrope", "default", None):
            dc.rope_scaling = None             # "mrope" only says how positions are assigned, the frequencies are plain
    if mt == "qwen2_moe":
        dc.num_experts, dc.top_k = cfg.num_experts, cfg.num_experts_per_tok
        dc.moe_intermediate_size = cfg.moe_intermediate_size
        dc.shared_expert_intermediate_size = cfg.shared_expert_intermediate_size
        dc.norm_topk_prob = bool(cfg.norm_topk_prob)
    return dc


def _lin(module):
    from .decoder import Lin
    from ..transformers.low_bit_linear import LowBitLinear
    if isinstance(module, LowBitLinear):
        return Lin(module.weight, module.bias)
    return Lin(module.weight.data, getattr(module, "bias", None))


def _split_lin_rows(module, sizes):
    """Split a (not yet device-tiled) merged linear into per-part ``Lin``s by dequantizing: only used for
    pre-merged architectures (phi3 gate_up_proj) where the engine wants gate and up separately."""
    from .decoder import Lin
    from ..transformers.low_bit_linear import LowBitLinear, FP4Params, dequant_weight
    if isinstance(module, LowBitLinear):
        w = dequant_weight(module.weight, torch.float32).cpu()
        qt = module.qtype
    else:
        w, qt = module.weight.data.float().cpu(), None
    outs, off = [], 0
    for s in sizes:
        part = w[off:off + s].contiguous()
        off += s
        if qt is not None and qt in Q.T16_QTYPES:
            p = FP4Params(part, qtype=qt).to("cpu").to(module.weight.data.device)
            outs.append(Lin(p))
        else:
            outs.append(Lin(part.to(module.weight.data.device)))
    return outs


def _text_parts(model):
    """(decoder stack module, text config) of a causal LM or of a multimodal wrapper around one."""
    cfg = model.config
    base = getattr(model, "model", None)
    if getattr(cfg, "model_type", None) in VL_ENGINE_TYPES:
        lm = getattr(base, "language_model", None) or getattr(model, "language_model", None)
        return lm, getattr(cfg, "text_config", None) or cfg.get_text_config()
    return base, cfg


def build_native_decoder(model):
    """Wrap the (already converted, already on the GPU) HF model's weights in a ``NativeDecoder``.  No copies for
    T16 weights: the engine reads the same device buffers as the LowBitLinear modules."""
    from .decoder import LayerWeights, NativeDecoder
    base, cfg = _text_parts(model)
    mt = cfg.model_type
    dc = decoder_config_from_hf(cfg)
    dev = base.embed_tokens.weight.device
    f16 = lambda t: t.detach().to(torch.float16).contiguous()  # noqa: E731
    layers = []
    for li, layer in enumerate(base.layers):
        att = layer.self_attn
        if hasattr(att, "qkv_proj"):
            qkv = _lin(att.qkv_proj)
        else:
            raise NotImplementedError("native engine needs merged qkv (optimize_model=True)")
        lw = LayerWeights(input_norm=f16(layer.input_layernorm.weight), qkv=qkv, o=_lin(att.o_proj),
                          post_norm=f16(layer.post_attention_layernorm.weight))
        mlp = layer.mlp
        if dc.num_experts:
            from .moe import MoEWeights
            lw.moe = MoEWeights.from_hf(mlp, dc, _lin)
        elif hasattr(mlp, "gate_up_proj"):
            lw.gate, lw.up = _split_lin_rows(mlp.gate_up_proj, (dc.intermediate_size, dc.intermediate_size))
            lw.down = _lin(mlp.down_proj)
        else:
            lw.gate, lw.up, lw.down = _lin(mlp.gate_proj), _lin(mlp.up_proj), _lin(mlp.down_proj)
            # (what the linears hold, not what the configuration says: a padded MLP is wider than an older config)
            dc.intermediate_size = max(dc.intermediate_size, lw.up.n)
        if mt == "gemma2":
            lw.post_attn_out_norm = f16(layer.post_attention_layernorm.weight)
            lw.pre_ffn_norm = f16(layer.pre_feedforward_layernorm.weight)
            lw.post_ffn_out_norm = f16(layer.post_feedforward_layernorm.weight)
            lw.post_norm = lw.pre_ffn_norm
            types_ = getattr(cfg, "layer_types", None)
            sliding = (types_[li] == "sliding_attention") if types_ else (li % 2 == 0)
            lw.window = dc.sliding_window if sliding else 0
        elif dc.sliding_window:
            lw.window = dc.sliding_window
        layers.append(lw)
    embed = base.embed_tokens.weight.detach()
    if embed.dtype != torch.float16:
        embed = embed.to(torch.float16)
    lm_head = _lin(model.lm_head)
    eng = NativeDecoder(dc, embed, layers, f16(base.norm.weight), lm_head, device=dev)
    return eng


def native_engine_supported(model) -> bool:
    """(the answer is cached on the model once an engine exists: the checks below walk every module, which is a
    millisecond per ``generate`` call on a 7B model)"""
    if getattr(model, "_native_engine", None) is not None:
        return True
    cfg = getattr(model, "config", None)
    if cfg is None or getattr(cfg, "model_type", None) not in NATIVE_ENGINE_TYPES | VL_ENGINE_TYPES:
        return False
    base, cfg = _text_parts(model)
    if base is None or not hasattr(base, "layers") or not hasattr(model, "lm_head"):
        return False
    first = base.layers[0].self_attn
    if not (hasattr(first, "qkv_proj")):
        return False
    from ..transformers.hybrid import HostResident
    if any(isinstance(m, HostResident) for m in model.modules()):
        return False          # CPU-GPU hybrid placement runs on the module path
    hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
    if hd not in (64, 96, 128, 256):
        return False
    if getattr(cfg, "model_type") == "granite" and float(getattr(cfg, "residual_multiplier", 1.0)) != 1.0:
        return False          # the engine adds projection outputs to the residual stream unscaled
    from .decoder import SUPPORTED_ROPE_TYPES
    rs = getattr(cfg, "rope_scaling", None) or getattr(cfg, "rope_parameters", None) or {}
    rope_type = rs.get("rope_type", rs.get("type"))
    if rope_type == "mrope":
        rope_type = "default"  # plain frequencies; the 3-axis position assignment is handled at prefill
    if rope_type not in SUPPORTED_ROPE_TYPES:
        return False          # e.g. dynamic NTK: HF's own forward handles it
    if getattr(cfg, "model_type") in ("mixtral", "qwen2_moe"):
        try:
            from . import moe  # noqa: F401
        except ImportError:
            return False
    return True


_PLAIN_GENERATE_KEYS = {"max_new_tokens", "min_new_tokens", "do_sample", "temperature", "top_k", "top_p",
                        "repetition_penalty", "eos_token_id", "pad_token_id", "streamer", "attention_mask",
                        "use_cache", "num_beams", "max_length", "generation_config", "return_dict_in_generate"}


def get_native_engine(model):
    eng = getattr(model, "_native_engine", None)
    if eng is None:
        eng = build_native_decoder(model)
        object.__setattr__(model, "_native_engine", eng)
    return eng


def native_generate(self, inputs=None, **kwargs):
    """``model.generate`` front door.  ``lookahead=n`` (or performance mode on a long prompt) selects prompt-lookup
    decoding, an attached draft model selects self-speculative decoding, plain greedy / sampling calls on the GPU
    go to the native engine, everything else to HuggingFace's generate."""
    from ..transformers.lookup import SPEC_ONLY_KWARGS, performance_mode_lookahead
    input_ids = kwargs.pop("input_ids", inputs)
    orig = self._hf_generate
    lookahead = kwargs.pop("lookahead", None)
    is_ids = isinstance(input_ids, torch.Tensor) and input_ids.dim() == 2
    if lookahead is None and is_ids:
        lookahead = performance_mode_lookahead(input_ids.shape)
    single = is_ids and input_ids.shape[0] == 1 and kwargs.get("num_beams", 1) in (None, 1)
    if lookahead and single and hasattr(self, "lookup_generate"):
        for k in SPEC_ONLY_KWARGS:
            kwargs.pop(k, None)
        return self.lookup_generate(input_ids, num_output_tokens=lookahead, **kwargs)
    kwargs.pop("max_matching_ngram_size", None)
    if getattr(self, "draft_model", None) is not None and single:
        return self.speculative_generate(input_ids, **kwargs)
    for k in SPEC_ONLY_KWARGS:
        kwargs.pop(k, None)
    vl_inputs = {k: kwargs[k] for k in _VL_INPUT_KEYS if kwargs.get(k) is not None}
    plain_keys = set(kwargs) - set(_VL_INPUT_KEYS) if getattr(self.config, "model_type", None) in VL_ENGINE_TYPES else set(kwargs)
    try_native = (is_ids
                  and next(self.parameters()).device.type == "cuda"
                  and plain_keys <= _PLAIN_GENERATE_KEYS and kwargs.get("num_beams", 1) == 1
                  and not kwargs.get("return_dict_in_generate", False) and native_engine_supported(self))
    am = kwargs.get("attention_mask")
    ragged = try_native and am is not None and not bool(am.all())
    if ragged and (vl_inputs or am.shape != input_ids.shape or not bool(am.any(-1).all())):
        try_native = False  # padded multimodal prompts / an all-padding row: HF path
    if not try_native:
        if input_ids is not None:
            kwargs["input_ids"] = input_ids
        return orig(**kwargs) if inputs is None else orig(inputs, **{k: v for k, v in kwargs.items() if k != "input_ids"})
    gc = kwargs.get("generation_config") or self.generation_config
    max_new = kwargs.get("max_new_tokens")
    if max_new is None:
        ml = kwargs.get("max_length", getattr(gc, "max_length", None) or 20)
        max_new = max(1, ml - input_ids.shape[1])
    from ..transformers.utils import engine_kv_mode, should_use_compresskv
    eng = get_native_engine(self)
    extra = {}
    if vl_inputs:
        embeds, pos = language_model_inputs(self, input_ids, vl_inputs, am)
        extra = dict(inputs_embeds=embeds, mrope_positions=pos if pos is not None and pos.dim() == 3 else None)
    # IPEX_LLM_COMPRESS_KV_CACHE=1: SnapKV selection at the end of every layer's prompt pass (engine prefill)
    eng.compress_kv = dict(window_size=32, max_capacity_prompt=1024, kernel_size=7, pooling="maxpool") \
        if should_use_compresskv(None, input_ids.shape[1]) and not ragged else None
    out = eng.generate(
        input_ids, max_new_tokens=max_new, min_new_tokens=kwargs.get("min_new_tokens", 0) or 0,
        do_sample=kwargs.get("do_sample", getattr(gc, "do_sample", False)) or False,
        temperature=kwargs.get("temperature", getattr(gc, "temperature", 1.0)) or 1.0,
        top_k=kwargs.get("top_k", getattr(gc, "top_k", 0)) or 0,
        top_p=kwargs.get("top_p", getattr(gc, "top_p", 1.0)) or 1.0,
        repetition_penalty=kwargs.get("repetition_penalty", getattr(gc, "repetition_penalty", 1.0)) or 1.0,
        eos_token_id=kwargs.get("eos_token_id", getattr(gc, "eos_token_id", None)),
        pad_token_id=kwargs.get("pad_token_id", getattr(gc, "pad_token_id", None)),
        streamer=kwargs.get("streamer"), kv_fp8=engine_kv_mode(), attention_mask=am if ragged else None, **extra)
    for attr in ("first_cost", "rest_cost_mean", "first_token_time", "last_token_time", "peak_memory"):
        object.__setattr__(self, attr, getattr(eng, attr))
    return out


_VL_INPUT_KEYS = ("pixel_values", "image_grid_thw", "pixel_values_videos", "video_grid_thw", "mm_token_type_ids",
                  "second_per_grid_ts")


class _LanguageModelCall(Exception):
    def __init__(self, kwargs):
        super().__init__("captured")
        self.kwargs = kwargs


@torch.no_grad()
def language_model_inputs(model, input_ids, vl_inputs, attention_mask=None):
    """What a multimodal wrapper hands to its language model for this prompt: (inputs_embeds [B, S, hidden] with the
    image / video features scattered over their placeholder tokens, position_ids -- [3, B, S] for Qwen2-VL's 3-axis
    rotary embedding).  The wrapper's own forward (vision tower, merger, placeholder mask, position index) runs
    unchanged up to the point where it calls the language model; that call is intercepted instead of executed, so
    nothing here re-implements family-specific preprocessing (reference models/qwen2_vl.py:58-185 re-binds the wrapper's
    forward for the same purpose)."""
    base = getattr(model, "model", model)
    lm = getattr(base, "language_model", None) or getattr(model, "language_model")
    original = lm.forward
    vl_inputs = dict(vl_inputs)
    if "mm_token_type_ids" not in vl_inputs and "mm_token_type_ids" in inspect.signature(base.forward).parameters:
        # the processor's token-type row (0 text, 1 image, 2 video), rebuilt from the placeholder ids
        cfg = model.config
        tt = torch.zeros_like(input_ids)
        if getattr(cfg, "image_token_id", None) is not None:
            tt = tt.masked_fill(input_ids == cfg.image_token_id, 1)
        if getattr(cfg, "video_token_id", None) is not None:
            tt = tt.masked_fill(input_ids == cfg.video_token_id, 2)
        vl_inputs["mm_token_type_ids"] = tt

    def intercept(*args, **kw):
        raise _LanguageModelCall(kw)
    lm.forward = intercept
    try:
        base(input_ids=input_ids, attention_mask=attention_mask, **vl_inputs)
    except _LanguageModelCall as call:
        return call.kwargs.get("inputs_embeds"), call.kwargs.get("position_ids")
    finally:
        lm.forward = original
    raise RuntimeError("the multimodal wrapper did not call its language model")


def _rms_forward_matches(m) -> bool:
    """Does ``rms_norm_forward`` compute what this module's own forward computes?  Checked once per class on a CPU
    copy of one instance with random weights: classes that only look like an RMSNorm (a weight that is never applied,
    a (1 + weight) scale under another name than Gemma's, an extra bias or gate) keep their own forward."""
    import copy
    try:
        probe = copy.deepcopy(m)
        if any(t.is_meta for t in list(probe.parameters()) + list(probe.buffers())):
            probe = probe.to_empty(device="cpu")
        probe = probe.float().cpu()
        with torch.no_grad():
            for t in list(probe.parameters()) + list(probe.buffers()):
                if t.is_floating_point():
                    t.uniform_(0.5, 1.5)
            x = torch.randn(2, 3, probe.weight.shape[0])
            ref = type(m).forward(probe, x)
            got = rms_norm_forward(probe, x)
        return isinstance(ref, torch.Tensor) and ref.shape == got.shape and torch.allclose(ref, got, atol=1e-4, rtol=1e-4)
    except Exception:
        return False


def _plain_gated_mlp(m) -> bool:
    """``down(act(gate(x)) * up(x))`` and nothing else: the class's forward takes one tensor and touches no attribute
    besides the three projections and the activation (sub-norms, gammas, expert gates keep HF's forward)."""
    import inspect
    import re
    try:
        fwd = type(m).forward
        params = list(inspect.signature(fwd).parameters.values())
        if len(params) != 2 or params[1].kind not in (params[1].POSITIONAL_ONLY, params[1].POSITIONAL_OR_KEYWORD):
            return False
        used = set(re.findall(r"self\.(\w+)", inspect.getsource(fwd)))
    except (OSError, TypeError, ValueError):
        return False
    return {"gate_proj", "up_proj", "down_proj"} <= used <= {"gate_proj", "up_proj", "down_proj", "act_fn"}


def optimize_post(model):
    """Bind the optimized forwards; install the native-engine generate front door."""
    from ..transformers.convert import convert_forward
    cfg = getattr(model, "config", None)
    seen = set()
    for m in model.modules():
        cls = m.__class__
        if cls in seen:
            continue
        name = cls.__name__
        if name.endswith("RMSNorm") and hasattr(m, "weight") and m.weight is not None and m.weight.dim() == 1:
            seen.add(cls)
            if _rms_forward_matches(m):
                convert_forward(model, cls, rms_norm_forward)
        elif cls is nn.LayerNorm:
            seen.add(cls)
            convert_forward(model, cls, layer_norm_forward)
        elif hasattr(m, "qkv_proj") and hasattr(m, "_merged_sizes") and hasattr(m, "o_proj"):
            seen.add(cls)
            convert_forward(model, cls, attention_forward)
        elif all(hasattr(m, p) for p in ("gate_proj", "up_proj", "down_proj", "act_fn")) and not list(
                getattr(m, "experts", [])) and not hasattr(m, "num_experts"):
            seen.add(cls)
            if _plain_gated_mlp(m):
                convert_forward(model, cls, mlp_forward)
    if getattr(cfg, "model_type", None) in ("rwkv", "rwkv5"):
        from .rwkv import bind_rwkv
        bind_rwkv(model)
    from .remote_code import bind_remote_code_forwards
    bind_remote_code_forwards(model)          # families that ship as remote code, recognised by their module protocol
    base = getattr(model, "model", None)
    if isinstance(base, nn.Module) and hasattr(base, "layers") and not getattr(base, "_kv_hook_installed", False) \
            and any(hasattr(getattr(l, "self_attn", None), "_merged_sizes") for l in base.layers):
        base.register_forward_pre_hook(kv_cache_pre_hook, with_kwargs=True)
        object.__setattr__(base, "_kv_hook_installed", True)
    if os.environ.get("BIGDL_HIP_ATTENTION", "1") != "0":
        register_hip_attention(model)
    install_generate_front_door(model)
    return model


def _raw_fp8_head_dim(cfg) -> bool:
    d = getattr(cfg, "head_dim", None) or (cfg.hidden_size // cfg.num_attention_heads
                                           if getattr(cfg, "num_attention_heads", 0) else 0)
    return d in (64, 96, 128, 256)


def kv_cache_pre_hook(module, args, kwargs):
    """Forward pre-hook of the base model: choose the KV cache class by policy (fp8 storage, SnapKV compression,
    else the block-preallocated cache) when the caller passed none or HF's empty default.  Reference: the cache
    selection at the top of each model forward, e.g. models/llama.py:73-94."""
    from transformers.cache_utils import DynamicCache
    from ..transformers.kv import make_kv_cache
    from ..transformers.utils import should_use_compresskv, use_quantize_kv_cache
    use_cache = kwargs.get("use_cache")
    if use_cache is None:
        use_cache = getattr(module.config, "use_cache", False)
    if not use_cache or module.training:
        return None
    pkv = kwargs.get("past_key_values")
    if pkv is not None and not (type(pkv) is DynamicCache and pkv.get_seq_length() == 0):
        return None
    ids = kwargs.get("input_ids", args[0] if args else None)
    emb = kwargs.get("inputs_embeds")
    seq = ids.shape[1] if ids is not None else (emb.shape[1] if emb is not None else 0)
    # fp16 models on the GPU whose attention goes through the registry entry below get the fp8 cache that is read as
    # bytes (no fp16 copy of the cache per step); BIGDL_FP8_RAW_CACHE=0 keeps the expanding one
    cfg = module.config
    p0 = next((p for p in module.parameters() if p.is_floating_point()), None)     # (low-bit weights are byte buffers)
    raw = (getattr(cfg, "_attn_implementation", None) == HIP_ATTENTION and p0 is not None and p0.device.type == "cuda"
           and p0.dtype == torch.float16 and _raw_fp8_head_dim(cfg) and os.environ.get("BIGDL_FP8_RAW_CACHE", "1") != "0")
    kwargs["past_key_values"] = make_kv_cache(use_quantize_kv_cache(), should_use_compresskv(prompt_len=seq), raw_fp8=raw)
    kwargs["use_cache"] = True
    return args, kwargs


def install_generate_front_door(model):
    if hasattr(model, "generate") and not hasattr(model, "_hf_generate"):
        object.__setattr__(model, "_hf_generate", model.generate)
        model.generate = types.MethodType(native_generate, model)
        if not hasattr(model, "lookup_generate"):
            from ..transformers.lookup import lookup_generate
            model.lookup_generate = types.MethodType(lookup_generate, model)
        if not hasattr(model, "native_engine"):
            # the NativeDecoder behind ``generate`` (built on first use from views of the module's weights), for callers
            # that drive prompt passes / decode steps / the KV cache themselves (examples/GPU/Applications/streaming-llm)
            model.native_engine = types.MethodType(get_native_engine, model)
    return model
