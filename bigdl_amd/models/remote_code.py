"""Forwards for families that only exist as remote code (``trust_remote_code=True`` modelling files that ship with the
checkpoint, not with ``transformers``).  Nothing here imports such a file: every function is written against the
attribute protocol of the module it replaces (the names the family's own modelling code gives its sub-modules and
buffers) and is bound by class name after conversion, when the modules' linears already are low-bit.

Capability map to the reference (python/llm/src/ipex_llm/transformers/models/):

* phixtral.py:44-104      ``phixtral_moeblock_forward`` / ``phixtral_mlp_forward``: phi-2 experts (fc1 / act / fc2)
* baichuan_m1.py:58-240   attention with a two-tap causal filter over the keys and values of neighbouring tokens
* chatglm.py:213-300      ChatGLM-6B (v1) attention: [seq, batch, hidden] layout, 2-D rotary positions, tuple caches
* qwen_vl.py:74-249       Qwen-VL (v1) text attention with a rotary table list, vision resampler, vision tower
* minicpmv.py:143-266     MiniCPM-V wrapper: generate front end that hands the merged embeddings to the language model
* internlm.py:235-381     InternLM-XComposer2: attention / MLP whose linears carry a partial LoRA for image tokens
* janus.py:24             Janus vision tower attention
* decilm.py:48            DeciLM: handled by the generic llama-layout binding on the module path (every layer merges its
                          own q / k / v, so per-layer kv head counts need nothing extra) and by
                          ``arch.py::convert_decilm`` on the native engine

Attention runs on the MFMA flash kernel (``ops/hip.py::attn_prefill``) for fp16 tensors on the GPU and on PyTorch's
``scaled_dot_product_attention`` otherwise.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# shared pieces
# ------------------------------------------------------------------------------------------------
def sdpa(q, k, v, mask=None, causal=False, scale=None):
    """q [B, Hq, S, D], k / v [B, Hkv, L, D] -> [B, S, Hq * D].  ``mask``: additive or boolean (True = attend),
    broadcastable to [B, Hq, S, L].  ``causal`` aligns the last query with the last key."""
    b, hq, s, d = q.shape
    scale = scale if scale is not None else 1.0 / math.sqrt(d)
    if q.is_cuda and q.dtype == torch.float16 and not torch.is_grad_enabled():
        from ..ops import hip
        if d in hip.PREFILL_HEAD_DIMS and all(t.stride(-1) == 1 for t in (q, k, v)) \
                and not any(t.stride(i) % 8 for t in (q, k, v) for i in (0, 1, 2)):
            m = mask
            if m is not None:
                if m.dtype == torch.bool:
                    m = torch.zeros(m.shape, dtype=torch.float16, device=m.device).masked_fill_(~m, float("-inf"))
                m = m.to(torch.float16)
                while m.dim() < 4:
                    m = m[None]
                m = m.expand(b, m.shape[1], s, k.shape[2]).contiguous() if m.shape[1] in (1, hq) else None
            if mask is None or m is not None:
                return hip.attn_prefill(q, k, v, scale=scale, causal=causal and mask is None, mask=m)
    g = hq // k.shape[1]
    if g > 1:
        k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
    if causal and mask is None and s > 1:
        L = k.shape[2]
        mask = torch.ones(s, L, dtype=torch.bool, device=q.device).tril(L - s)
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=scale)
    return out.transpose(1, 2).reshape(b, s, hq * d)


def rotate_half(x):
    return torch.cat([-x[..., x.shape[-1] // 2:], x[..., : x.shape[-1] // 2]], -1)


def rope_tables(positions, dim, base=10000.0, dtype=torch.float32):
    """cos / sin [..., dim] for integer ``positions`` [...] (half-split convention: the table repeats over both halves)"""
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, device=positions.device, dtype=torch.float32) / dim))
    ang = positions[..., None].float() * inv
    ang = torch.cat([ang, ang], -1)
    return ang.cos().to(dtype), ang.sin().to(dtype)


# ------------------------------------------------------------------------------------------------
# phixtral: mixture of phi-2 MLPs
# ------------------------------------------------------------------------------------------------
def phixtral_mlp_forward(self, x):
    return self.fc2(self.act(self.fc1(x)))


def phixtral_moeblock_forward(self, hidden_states):
    """Protocol: ``self.gate`` (router linear), ``self.mlp`` (list of experts with fc1 / act / fc2),
    ``self.num_experts_per_tok``.  Softmax over all experts, top-k renormalized, weighted sum of the chosen experts.
    Decode-sized inputs (one row) run exactly the k chosen experts; larger inputs are grouped by expert so that each
    expert runs once over its rows."""
    shape = hidden_states.shape
    x = hidden_states.reshape(-1, shape[-1])
    probs = F.softmax(self.gate(x), dim=-1, dtype=torch.float32)
    w, idx = torch.topk(probs, self.num_experts_per_tok, dim=-1)
    w = (w / w.sum(-1, keepdim=True)).to(x.dtype)
    if x.shape[0] == 1:
        out = None
        for e, we in zip(idx[0].tolist(), w[0]):
            y = self.mlp[e](x) * we
            out = y if out is None else out + y
        return out.reshape(shape)
    out = torch.zeros_like(x)
    flat = idx.reshape(-1)
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=len(self.mlp)).tolist()
    tok = torch.div(order, idx.shape[1], rounding_mode="floor")
    wf = w.reshape(-1)[order]
    at = 0
    for e, c in enumerate(counts):
        if c:
            rows = tok[at:at + c]
            out.index_add_(0, rows, self.mlp[e](x[rows]) * wf[at:at + c, None])
            at += c
    return out.reshape(shape)


# ------------------------------------------------------------------------------------------------
# Baichuan-M1: keys and values are filtered over neighbouring tokens before attention
# ------------------------------------------------------------------------------------------------
def two_tap_filter(u, taps, last=None):
    """u [B, S, H, D]; taps [1, 1, H, 1, 2] -> taps[..., 0] * u[t - 1] + taps[..., 1] * u[t] (u[-1] = ``last`` or 0)"""
    c0, c1 = taps[0, 0, :, 0, 0].view(1, 1, -1, 1), taps[0, 0, :, 0, 1].view(1, 1, -1, 1)
    prev = torch.cat([torch.zeros_like(u[:, :1]) if last is None else last, u[:, :-1]], 1)
    return c0.to(u.dtype) * prev + c1.to(u.dtype) * u


def baichuan_m1_attention_forward(self, hidden_states, attention_mask=None, position_ids=None, seqlens=None,
                                  past_key_value=None, output_attentions=False, use_cache=False, cache_position=None,
                                  position_embeddings=None, **kwargs):
    """Protocol: ``W_pack`` (packed q | k | v), ``conv_k`` / ``conv_v`` [1, 1, Hkv, 1, 2], ``o_proj``, ``num_heads``,
    ``num_key_value_heads``, ``head_dim``, ``layer_idx``, ``rotary_emb.inv_freq``; the cache is a ``Cache`` with
    ``update(k, v, layer_idx)``.  The unfiltered key / value of the newest token stay on the module (``last_k`` /
    ``last_v``): the filter of the next step needs them, the cache holds filtered rows only."""
    assert seqlens is None, "packed sequences are not supported"
    b, s, _ = hidden_states.shape
    hq, hkv, d = self.num_heads, self.num_key_value_heads, self.head_dim
    qkv = self.W_pack(hidden_states).view(b, s, hq + 2 * hkv, d)
    q, k, v = qkv.split([hq, hkv, hkv], dim=2)
    fresh = past_key_value is None or past_key_value.get_seq_length(self.layer_idx) == 0
    last_k, last_v = (None, None) if fresh else (self.last_k, self.last_v)
    self.last_k, self.last_v = k[:, -1:], v[:, -1:]
    k, v = two_tap_filter(k, self.conv_k, last_k), two_tap_filter(v, self.conv_v, last_v)
    if position_ids is None:
        past = 0 if fresh else past_key_value.get_seq_length(self.layer_idx)
        position_ids = torch.arange(past, past + s, device=hidden_states.device)[None]
    inv = self.rotary_emb.inv_freq.to(hidden_states.device).float()
    ang = position_ids[..., None].float() * inv                                  # [B, S, D/2]
    cos, sin = torch.cat([ang.cos()] * 2, -1)[:, :, None].to(q.dtype), torch.cat([ang.sin()] * 2, -1)[:, :, None].to(q.dtype)
    q = (q * cos + rotate_half(q) * sin).transpose(1, 2)
    k = (k * cos + rotate_half(k) * sin).transpose(1, 2)
    v = v.transpose(1, 2)
    if past_key_value is not None:
        k, v = past_key_value.update(k, v, self.layer_idx, None)
    out = sdpa(q, k, v, mask=attention_mask, causal=attention_mask is None and s > 1)
    return self.o_proj(out.to(hidden_states.dtype)), None, past_key_value


# ------------------------------------------------------------------------------------------------
# ChatGLM-6B (v1)
# ------------------------------------------------------------------------------------------------
def chatglm_v1_attention_forward(self, hidden_states, position_ids, attention_mask, layer_id, layer_past=None,
                                 use_cache=False, output_attentions=False):
    """hidden_states [S, B, hidden]; position_ids [B, 2, S] (token positions, block positions) when
    ``position_encoding_2d``; attention_mask boolean [B, 1, S, L] with True = masked; cache = (k, v) as [L, B, H, D].
    Protocol: ``query_key_value`` (per head q | k | v), ``dense``, ``num_attention_heads_per_partition``,
    ``hidden_size_per_attention_head``, ``position_encoding_2d``, ``rotary_emb.inv_freq`` or ``rotary_emb.base``.
    The family scales q by 1 / (sqrt(d) * (layer_id + 1)) and the scores by (layer_id + 1): the product is the
    plain 1 / sqrt(d), which is what runs here."""
    s, b, _ = hidden_states.shape
    h, d = self.num_attention_heads_per_partition, self.hidden_size_per_attention_head
    mixed = self.query_key_value(hidden_states).view(s, b, h, 3 * d)
    q, k, v = mixed.split(d, dim=-1)                                             # [S, B, H, D]
    base = float(getattr(self.rotary_emb, "base", 10000.0))

    def rot(x, pos, dim):                                                        # pos [B, S]
        cos, sin = rope_tables(pos.transpose(0, 1), dim, base, x.dtype)          # [S, B, dim]
        return x * cos[:, :, None] + rotate_half(x) * sin[:, :, None]

    if self.position_encoding_2d:
        p, bp = position_ids[:, 0, :], position_ids[:, 1, :]
        half = d // 2
        q = torch.cat([rot(q[..., :half], p, half), rot(q[..., half:], bp, half)], -1)
        k = torch.cat([rot(k[..., :half], p, half), rot(k[..., half:], bp, half)], -1)
    else:
        q, k = rot(q, position_ids, d), rot(k, position_ids, d)
    if layer_past is not None:
        k, v = torch.cat([layer_past[0], k], 0), torch.cat([layer_past[1], v], 0)
    present = (k, v) if use_cache else None
    mask = None
    if attention_mask is not None and attention_mask.numel() > 1:
        mask = ~attention_mask if attention_mask.dtype == torch.bool else attention_mask
    out = sdpa(q.permute(1, 2, 0, 3), k.permute(1, 2, 0, 3), v.permute(1, 2, 0, 3), mask=mask,
               causal=mask is None and s > 1)                                    # [B, S, H * D]
    out = self.dense(out.transpose(0, 1).contiguous())
    return (out, present, None) if output_attentions else (out, present)


# ------------------------------------------------------------------------------------------------
# Qwen-VL (v1)
# ------------------------------------------------------------------------------------------------
def qwen_vl_attention_forward(self, hidden_states, rotary_pos_emb=None, registered_causal_mask=None, layer_past=None,
                              attention_mask=None, head_mask=None, encoder_hidden_states=None,
                              encoder_attention_mask=None, output_attentions=False, use_cache=False):
    """Text attention of Qwen-VL: packed ``c_attn`` with bias, ``c_proj``; ``rotary_pos_emb`` is the model's list
    [cos, sin] of [1, L, 1, D] tables (the newest ``S`` rows apply); cache = (k, v) as [B, L, H, D]; log-n scaling of
    q past the training length when ``use_logn_attn``."""
    b, s, _ = hidden_states.shape
    h, d = self.num_heads, self.head_dim
    q, k, v = self.c_attn(hidden_states).split(self.split_size, dim=2)
    q, k, v = q.view(b, s, h, d), k.view(b, s, h, d), v.view(b, s, h, d)
    if rotary_pos_emb is not None:
        cos, sin = (t[:, -s:].to(q.dtype) for t in rotary_pos_emb)
        rd = cos.shape[-1]
        q = torch.cat([q[..., :rd] * cos + rotate_half(q[..., :rd]) * sin, q[..., rd:]], -1)
        k = torch.cat([k[..., :rd] * cos + rotate_half(k[..., :rd]) * sin, k[..., rd:]], -1)
    if layer_past is not None:
        k, v = torch.cat([layer_past[0], k], 1), torch.cat([layer_past[1], v], 1)
    present = (k, v) if use_cache else None
    L = k.shape[1]
    if getattr(self, "use_logn_attn", False) and not self.training:
        logn = self.logn_tensor.to(q.device, q.dtype)[:, L - s:L]
        q = q * logn
    mask = attention_mask if attention_mask is not None and attention_mask.numel() > 1 else None
    if mask is not None and s > 1:               # the model passes a padding mask only: add the causal structure
        causal = torch.ones(s, L, dtype=torch.bool, device=q.device).tril(L - s)
        mask = mask.to(q.dtype) + torch.zeros(s, L, dtype=q.dtype, device=q.device).masked_fill_(~causal, float("-inf"))
    out = sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask=mask, causal=mask is None and s > 1)
    out = self.c_proj(out)
    return (out, present, None) if output_attentions else (out, present)


def _abs_pos(table, tgt_len):
    """Learned 2-D position table [src * src, C] resized (bicubic) to ``tgt_len`` = tgt * tgt positions."""
    src, tgt = int(math.sqrt(table.shape[0])), int(math.sqrt(tgt_len))
    if src == tgt:
        return table
    t = table.float().reshape(1, src, src, -1).permute(0, 3, 1, 2)
    t = F.interpolate(t, size=(tgt, tgt), mode="bicubic", align_corners=False)
    return t.permute(0, 2, 3, 1).flatten(0, 2).to(table.dtype)


def qwen_vl_resampler_forward(self, x, attn_mask=None):
    """Cross-attention pooling of the image tokens onto ``self.query`` ([n_queries, C]); ``self.attn`` is an
    ``nn.MultiheadAttention`` in [L, N, C] layout."""
    pos = _abs_pos(self.pos_embed, x.size(1))
    x = self.ln_kv(self.kv_proj(x)).permute(1, 0, 2)
    n = x.shape[1]
    q = self.ln_q(self.query)
    out = self.attn(q.unsqueeze(1).repeat(1, n, 1) + self.pos_embed.unsqueeze(1), x + pos.unsqueeze(1), x,
                    attn_mask=attn_mask)[0]
    return out.permute(1, 0, 2)


def qwen_vl_vision_transformer_forward(self, x):
    p = next(self.transformer.parameters())
    x = self.conv1(x.to(device=p.device, dtype=p.dtype if p.is_floating_point() else x.dtype))
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    x = self.ln_pre(x + _abs_pos(self.positional_embedding, x.size(1)))
    x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
    return self.ln_post(self.attn_pool(x)) @ self.proj


def vision_mha_forward(self, x, attn_mask=None):
    """Self-attention of a CLIP-style vision block in [L, N, C] layout (Qwen-VL's ``VisualAttention``: ``in_proj``,
    ``out_proj``, ``num_attention_heads_per_partition``, ``hidden_size_per_attention_head``) on the flash kernel."""
    L, n, _ = x.shape
    h, d = self.num_attention_heads_per_partition, self.hidden_size_per_attention_head
    q, k, v = self.in_proj(x).view(L, n, h, 3 * d).split(d, dim=-1)
    out = sdpa(q.permute(1, 2, 0, 3), k.permute(1, 2, 0, 3), v.permute(1, 2, 0, 3), mask=attn_mask)
    return self.out_proj(out.transpose(0, 1).contiguous())


# ------------------------------------------------------------------------------------------------
# Janus vision tower
# ------------------------------------------------------------------------------------------------
def janus_vision_attention_forward(self, x):
    """Protocol: ``qkv`` ([B, N, C] -> 3C), ``num_heads``, ``head_dim``, optional ``q_norm`` / ``k_norm``, ``proj``,
    ``proj_drop``."""
    b, n, c = x.shape
    qkv = self.qkv(x).view(b, n, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    if getattr(self, "q_norm", None) is not None:
        q, k = self.q_norm(q), self.k_norm(k)
    out = sdpa(q.contiguous(), k.contiguous(), v.contiguous())
    out = self.proj(out)
    return self.proj_drop(out) if hasattr(self, "proj_drop") else out


# ------------------------------------------------------------------------------------------------
# InternLM-XComposer2: linears with a partial LoRA that only image tokens go through
# ------------------------------------------------------------------------------------------------
def lift_plora(module):
    """Before conversion: move the partial-LoRA parts of every ``PLoRA`` child (a linear subclass holding ``Plora_A`` /
    ``Plora_B`` / ``lora_scaling`` / ``lora_dropout``) onto the parent as ``<child>_Plora_A`` ..., and put a plain
    ``nn.Linear`` with the same weight in the child's place, so that the converter sees an ordinary linear and makes
    it low-bit while the small adapter matrices stay dense."""
    from torch import nn
    if type(module).__name__ not in ("InternLM2Attention", "InternLM2MLP"):
        return
    for name, child in list(module.named_children()):
        if not hasattr(child, "Plora_A"):
            continue
        child.Plora_A._bigdl_keep_dense = child.Plora_B._bigdl_keep_dense = True
        setattr(module, name + "_Plora_A", child.Plora_A)
        setattr(module, name + "_Plora_B", child.Plora_B)
        module.__dict__[name + "_lora_scaling"] = float(getattr(child, "lora_scaling", 1.0))
        drop = getattr(child, "lora_dropout", None)
        if drop is not None:
            setattr(module, name + "_lora_dropout", drop)
        plain = nn.Linear(child.in_features, child.out_features, bias=child.bias is not None, device="meta")
        plain.weight, plain.bias = child.weight, child.bias
        setattr(module, name, plain)


def plora_apply(parent, name, x, im_mask=None):
    """``y = linear(x)``, plus ``B(A(x)) * scaling`` on the rows ``im_mask`` marks (image tokens).  ``im_mask``: boolean
    [B, S], a list of (start, end) row ranges of a one-sequence batch, or None / empty."""
    y = getattr(parent, name)(x)
    a = getattr(parent, name + "_Plora_A", None)
    if a is None or im_mask is None or (not isinstance(im_mask, torch.Tensor) and len(im_mask) == 0):
        return y
    if not isinstance(im_mask, torch.Tensor):
        m = torch.zeros(x.shape[:-1], dtype=torch.bool, device=x.device)
        for lo, hi in im_mask:
            m[0, lo:hi] = True
        im_mask = m
    im_mask = im_mask.to(x.device).bool()
    if im_mask.shape != x.shape[:-1] or not bool(im_mask.any()):
        return y
    b = getattr(parent, name + "_Plora_B")
    drop = getattr(parent, name + "_lora_dropout", None)
    part = x[im_mask]
    part = drop(part) if drop is not None else part
    delta = b(a(part.to(a.weight.dtype))) * getattr(parent, name + "_lora_scaling")
    y = y.clone()
    y[im_mask] = y[im_mask] + delta.to(y.dtype)
    return y


def xcomposer2_attention_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                                 output_attentions=False, use_cache=False, im_mask=None, **kwargs):
    """Protocol: ``wqkv`` / ``wo`` PLoRA linears, grouped packing [kv_head, (group q heads | k | v), d] as in
    InternLM2, ``num_heads``, ``num_key_value_heads``, ``head_dim``, ``rotary_emb`` giving (cos, sin) [L, D];
    cache = (k, v) as [B, H, L, D]."""
    b, s, _ = hidden_states.shape
    hq, hkv, d = self.num_heads, self.num_key_value_heads, self.head_dim
    g = hq // hkv
    qkv = plora_apply(self, "wqkv", hidden_states, im_mask).view(b, s, hkv, g + 2, d)
    q = qkv[:, :, :, :g].reshape(b, s, hq, d).transpose(1, 2)
    k, v = qkv[:, :, :, g].transpose(1, 2), qkv[:, :, :, g + 1].transpose(1, 2)
    past = 0 if past_key_value is None else past_key_value[0].shape[2]
    if position_ids is None:
        position_ids = torch.arange(past, past + s, device=hidden_states.device)[None]
    base = float(getattr(self.rotary_emb, "base", 10000.0))
    cos, sin = rope_tables(position_ids, d, base, q.dtype)                       # [B, S, D]
    cos, sin = cos[:, None], sin[:, None]
    q, k = q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin
    if past_key_value is not None:
        k, v = torch.cat([past_key_value[0], k], 2), torch.cat([past_key_value[1], v], 2)
    present = (k, v) if use_cache else None
    out = sdpa(q, k, v, mask=attention_mask, causal=attention_mask is None and s > 1)
    return plora_apply(self, "wo", out.to(hidden_states.dtype), im_mask), None, present


def xcomposer2_mlp_forward(self, x, im_mask=None):
    return plora_apply(self, "w2", F.silu(plora_apply(self, "w1", x, im_mask)) * plora_apply(self, "w3", x, im_mask), im_mask)


# ------------------------------------------------------------------------------------------------
# MiniCPM-V wrapper
# ------------------------------------------------------------------------------------------------
def minicpmv_generate_wrapper(origin_generate):
    """MiniCPM-V's ``generate`` / ``chat`` build ``inputs_embeds`` from the image features and then call the language
    model's ``generate`` with whatever the caller passed.  The wrapper turns a repetition penalty into the in-place
    device kernel by way of a logits processor and defaults to greedy decoding when no sampling argument was given (reference
    minicpmv.py:143-266: patched ``generate`` + fused repetition-penalty hook)."""
    def generate(self, *args, **kwargs):
        penalty = kwargs.pop("repetition_penalty", None)
        if penalty is not None and penalty != 1.0:
            from transformers import LogitsProcessorList
            procs = kwargs.get("logits_processor") or LogitsProcessorList()
            procs.append(RepetitionPenaltyInplace(penalty))
            kwargs["logits_processor"] = procs
        kwargs.setdefault("do_sample", False)
        return origin_generate(self, *args, **kwargs)
    generate._bigdl_wrapped = True
    return generate


class RepetitionPenaltyInplace:
    """Logits processor: the HF repetition penalty computed in place by one kernel on the GPU
    (``ops/hip.py::repetition_penalty_inplace``), by the usual gather / scatter elsewhere."""

    def __init__(self, penalty: float):
        self.penalty = float(penalty)

    def __call__(self, input_ids, scores):
        if scores.is_cuda and scores.dtype == torch.float32 and scores.is_contiguous():
            from ..ops import hip
            hip.repetition_penalty_inplace(scores, input_ids, self.penalty)
            return scores
        picked = scores.gather(1, input_ids)
        picked = torch.where(picked < 0, picked * self.penalty, picked / self.penalty)
        return scores.scatter(1, input_ids, picked)


# ------------------------------------------------------------------------------------------------
# binding
# ------------------------------------------------------------------------------------------------
def _bind(model, predicate, forward):
    import types
    n = 0
    for m in model.modules():
        if predicate(m):
            m.forward = types.MethodType(forward, m)
            n += 1
    return n


def _has(m, *names):
    return all(hasattr(m, n) for n in names)


def pre_process_remote_code(model):
    """Structural changes that have to precede the low-bit conversion."""
    for m in list(model.modules()):
        lift_plora(m)


def bind_remote_code_forwards(model) -> int:
    """Replace the forwards of the remote-code modules found in ``model`` (recognised by class name plus the attributes
    the forward needs).  Returns how many modules were rebound."""
    n = 0
    n += _bind(model, lambda m: type(m).__name__ == "MoE" and _has(m, "gate", "mlp", "num_experts_per_tok"),
               phixtral_moeblock_forward)
    n += _bind(model, lambda m: type(m).__name__ == "BaichuanAttention" and _has(m, "W_pack", "conv_k", "conv_v", "o_proj"),
               baichuan_m1_attention_forward)
    n += _bind(model, lambda m: type(m).__name__ == "SelfAttention" and _has(m, "query_key_value", "dense", "position_encoding_2d"),
               chatglm_v1_attention_forward)
    n += _bind(model, lambda m: type(m).__name__ == "QWenAttention" and _has(m, "c_attn", "c_proj", "split_size")
               and getattr(getattr(model, "config", None), "visual", None) is not None, qwen_vl_attention_forward)
    n += _bind(model, lambda m: type(m).__name__ == "Resampler" and _has(m, "kv_proj", "ln_kv", "ln_q", "query", "attn", "pos_embed"),
               qwen_vl_resampler_forward)
    n += _bind(model, lambda m: type(m).__name__ == "VisionTransformer" and _has(m, "conv1", "ln_pre", "transformer", "attn_pool", "proj"),
               qwen_vl_vision_transformer_forward)
    n += _bind(model, lambda m: type(m).__name__ == "VisualAttention" and _has(m, "in_proj", "out_proj"), vision_mha_forward)
    n += _bind(model, lambda m: type(m).__name__ == "Attention" and _has(m, "qkv", "proj", "num_heads", "head_dim")
               and "janus" in str(getattr(getattr(model, "config", None), "model_type", "")).lower() + type(model).__name__.lower(),
               janus_vision_attention_forward)
    n += _bind(model, lambda m: type(m).__name__ == "InternLM2Attention" and _has(m, "wqkv", "wo", "wqkv_Plora_A"),
               xcomposer2_attention_forward)
    n += _bind(model, lambda m: type(m).__name__ == "InternLM2MLP" and _has(m, "w1", "w2", "w3", "w1_Plora_A"),
               xcomposer2_mlp_forward)
    if type(model).__name__ == "MiniCPMV" and hasattr(model, "llm") and not getattr(type(model).generate, "_bigdl_wrapped", False):
        import types
        model.generate = types.MethodType(minicpmv_generate_wrapper(type(model).generate), model)
        n += 1
    return n
