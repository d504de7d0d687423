"""Decoder families that HuggingFace only ships as remote code, served by the native engine.

The reference optimizes these by rebinding the forwards of the checkpoints' own ``modeling_*.py``
(python/llm/src/ipex_llm/transformers/models/{chatglm2,chatglm4,baichuan,qwen,internlm,minicpm,aquila}.py, bound in
convert.py:1301-2025), which needs ``trust_remote_code`` and the files next to the weights.  Here a family is a
*weight-name map*: its checkpoint tensors are renamed / re-split into one canonical llama-like state dict plus a
``DecoderConfig`` that switches on what the family needs (qkv bias, partial or interleaved rotary, multi-query groups,
embedding / residual / logit scales).  The canonical form is then either quantized into the native engine (GPU) or
run by the plain PyTorch forward in this file (CPU; also the numerics reference of the tests).  No remote code runs.

Canonical tensor names (all ``[out, in]`` like nn.Linear):
``embed``, ``final_norm``, ``lm_head``, and per layer ``L{i}.input_norm``, ``L{i}.qkv`` (+ ``.bias``; rows q | k | v),
``L{i}.o`` (+ ``.bias``), ``L{i}.post_norm``, ``L{i}.gate``, ``L{i}.up``, ``L{i}.down``.
"""
from __future__ import annotations

import json
import math
import os
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .. import qtypes as Q
from .decoder import DecoderConfig, alibi_slopes, compute_inv_freq

StateDict = Dict[str, torch.Tensor]


def _get(cfg: dict, *names, default=None):
    for n in names:
        if n in cfg and cfg[n] is not None:
            return cfg[n]
    return default


# ------------------------------------------------------------------------------------------------
# family converters: (config dict, checkpoint state dict) -> (DecoderConfig, canonical state dict)
# ------------------------------------------------------------------------------------------------
def _llama_keys(cfg: dict, sd: StateDict, dc: DecoderConfig, *, qkv: Optional[str] = None, o_scale=1.0, down_scale=1.0):
    """Checkpoints that already use llama's module names (baichuan-7B with a packed W_pack, internlm, minicpm,
    aquila, yi)."""
    out = {"embed": sd["model.embed_tokens.weight"], "final_norm": sd["model.norm.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    for i in range(dc.num_layers):
        p = f"model.layers.{i}."
        if qkv is not None:
            out[f"L{i}.qkv"] = sd[p + f"self_attn.{qkv}.weight"]
            if p + f"self_attn.{qkv}.bias" in sd:
                out[f"L{i}.qkv.bias"] = sd[p + f"self_attn.{qkv}.bias"]
        else:
            out[f"L{i}.qkv"] = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
            if p + "self_attn.q_proj.bias" in sd:
                out[f"L{i}.qkv.bias"] = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        out[f"L{i}.o"] = sd[p + "self_attn.o_proj.weight"] * o_scale
        if p + "self_attn.o_proj.bias" in sd:
            out[f"L{i}.o.bias"] = sd[p + "self_attn.o_proj.bias"] * o_scale
        out[f"L{i}.input_norm"] = sd[p + "input_layernorm.weight"]
        out[f"L{i}.post_norm"] = sd[p + "post_attention_layernorm.weight"]
        out[f"L{i}.gate"] = sd[p + "mlp.gate_proj.weight"]
        out[f"L{i}.up"] = sd[p + "mlp.up_proj.weight"]
        out[f"L{i}.down"] = sd[p + "mlp.down_proj.weight"] * down_scale
    return out


def _base_config(cfg: dict, model_type: str, **over) -> DecoderConfig:
    heads = _get(cfg, "num_attention_heads", "n_head", "n_heads", default=over.get("num_heads"))
    hidden = _get(cfg, "hidden_size", "n_embd", "d_model", default=over.get("hidden_size"))
    rp = cfg.get("rope_parameters") or {}          # transformers >= 5 writes theta / scaling into one dict
    if rp and "rope_theta" not in cfg:
        cfg = dict(cfg, rope_theta=rp.get("rope_theta", 10000.0))
        if rp.get("rope_type", "default") != "default" and not cfg.get("rope_scaling"):
            cfg["rope_scaling"] = {k: v for k, v in rp.items() if k != "rope_theta"}
    kw = dict(vocab_size=_get(cfg, "vocab_size", "padded_vocab_size"), hidden_size=hidden,
              intermediate_size=_get(cfg, "intermediate_size", "ffn_hidden_size", default=0),
              num_layers=_get(cfg, "num_hidden_layers", "num_layers", "n_layer"), num_heads=heads,
              num_kv_heads=_get(cfg, "num_key_value_heads", default=heads),
              head_dim=_get(cfg, "head_dim", "kv_channels", default=hidden // heads),
              rms_eps=_get(cfg, "rms_norm_eps", "layernorm_epsilon", "layer_norm_epsilon", default=1e-6),
              rope_theta=float(_get(cfg, "rope_theta", "rotary_emb_base", default=10000.0)),
              rope_scaling=_get(cfg, "rope_scaling"),
              max_position=_get(cfg, "max_position_embeddings", "seq_length", "model_max_length", default=4096),
              tie_word_embeddings=bool(_get(cfg, "tie_word_embeddings", default=False)), model_type=model_type)
    kw.update(over)
    return DecoderConfig(**kw)


def convert_chatglm(cfg: dict, sd: StateDict):
    """ChatGLM2 / ChatGLM3 / GLM-4 (``model_type: chatglm`` with RMSNorm; reference models/chatglm2.py, chatglm4.py).
    Rotary embedding on the first half of each head, interleaved pairs; multi-query groups; qkv bias; the MLP's
    ``dense_h_to_4h`` holds gate and up stacked."""
    if not _get(cfg, "rmsnorm", default=True) or _get(cfg, "position_encoding_2d", default=False):
        raise NotImplementedError("ChatGLM-6B (v1: LayerNorm, 2-D rotary, GELU) is not covered by the native engine")
    heads = cfg["num_attention_heads"]
    d = _get(cfg, "kv_channels", default=cfg["hidden_size"] // heads)
    mqa = bool(_get(cfg, "multi_query_attention", default=False))
    nkv = _get(cfg, "multi_query_group_num", default=heads) if mqa else heads
    dc = _base_config(cfg, "chatglm", num_kv_heads=nkv, head_dim=d, rot_dim=d // 2, rope_interleaved=True,
                      rope_theta=10000.0 * float(_get(cfg, "rope_ratio", default=1.0)),
                      intermediate_size=cfg["ffn_hidden_size"], vocab_size=_get(cfg, "padded_vocab_size", "vocab_size"))
    ffn = dc.intermediate_size
    out = {"embed": sd["transformer.embedding.word_embeddings.weight"],
           "final_norm": sd["transformer.encoder.final_layernorm.weight"],
           "lm_head": sd["transformer.output_layer.weight"]}
    for i in range(dc.num_layers):
        p = f"transformer.encoder.layers.{i}."
        w = sd[p + "self_attention.query_key_value.weight"]
        b = sd.get(p + "self_attention.query_key_value.bias")
        if not mqa:
            # [heads, (q, k, v), d] per head -> q | k | v
            w = w.view(heads, 3, d, -1).transpose(0, 1).reshape(3 * heads * d, -1)
            b = None if b is None else b.view(heads, 3, d).transpose(0, 1).reshape(-1)
        out[f"L{i}.qkv"] = w
        if b is not None:
            out[f"L{i}.qkv.bias"] = b
        out[f"L{i}.o"] = sd[p + "self_attention.dense.weight"]
        out[f"L{i}.input_norm"] = sd[p + "input_layernorm.weight"]
        out[f"L{i}.post_norm"] = sd[p + "post_attention_layernorm.weight"]
        h4 = sd[p + "mlp.dense_h_to_4h.weight"]
        out[f"L{i}.gate"], out[f"L{i}.up"] = h4[:ffn], h4[ffn:]
        out[f"L{i}.down"] = sd[p + "mlp.dense_4h_to_h.weight"]
    return dc, out


def convert_baichuan(cfg: dict, sd: StateDict):
    """Baichuan-7B / Baichuan2-7B (reference models/baichuan.py:206-286): llama with a packed ``W_pack``; Baichuan2's
    ``NormHead`` normalizes the lm_head rows at inference, folded into the weight here (reference
    convert.py:909-923).  The 13B models (``model_max_length`` instead of ``max_position_embeddings`` in their config)
    use ALiBi instead of rotary embeddings (reference models/baichuan.py:289-372)."""
    dc = _base_config(cfg, "baichuan", alibi="max_position_embeddings" not in cfg)
    out = _llama_keys(cfg, sd, dc, qkv="W_pack")
    if cfg.get("vocab_size", 0) >= 125696:           # Baichuan2 vocabulary: NormHead
        out["lm_head"] = F.normalize(out["lm_head"].float(), dim=-1).to(out["lm_head"].dtype)
    return dc, out


def convert_qwen(cfg: dict, sd: StateDict):
    """Qwen (v1) 1.8B-72B (reference models/qwen.py): packed ``c_attn`` with bias, RMSNorm, ``c_proj(w1(x) * silu(w2(x)))``."""
    dc = _base_config(cfg, "qwen", intermediate_size=cfg["intermediate_size"] // 2,
                      head_dim=_get(cfg, "kv_channels", default=cfg["hidden_size"] // cfg["num_attention_heads"]))
    out = {"embed": sd["transformer.wte.weight"], "final_norm": sd["transformer.ln_f.weight"], "lm_head": sd["lm_head.weight"]}
    for i in range(dc.num_layers):
        p = f"transformer.h.{i}."
        out[f"L{i}.qkv"] = sd[p + "attn.c_attn.weight"]
        out[f"L{i}.qkv.bias"] = sd[p + "attn.c_attn.bias"]
        out[f"L{i}.o"] = sd[p + "attn.c_proj.weight"]
        out[f"L{i}.input_norm"] = sd[p + "ln_1.weight"]
        out[f"L{i}.post_norm"] = sd[p + "ln_2.weight"]
        out[f"L{i}.gate"] = sd[p + "mlp.w2.weight"]
        out[f"L{i}.up"] = sd[p + "mlp.w1.weight"]
        out[f"L{i}.down"] = sd[p + "mlp.c_proj.weight"]
    return dc, out


def convert_internlm(cfg: dict, sd: StateDict):
    """InternLM (v1; reference models/internlm.py:56-113): llama with biases on q/k/v/o."""
    dc = _base_config(cfg, "internlm")
    return dc, _llama_keys(cfg, sd, dc)


def convert_internlm2(cfg: dict, sd: StateDict):
    """InternLM2 (reference models/internlm.py:116-183): ``wqkv`` stores, per key/value head, its group of query
    heads followed by the key and the value head."""
    dc = _base_config(cfg, "internlm2")
    hq, hkv, d = dc.num_heads, dc.num_kv_heads, dc.head_dim
    g = hq // hkv
    out = {"embed": sd["model.tok_embeddings.weight"], "final_norm": sd["model.norm.weight"], "lm_head": sd["output.weight"]}
    for i in range(dc.num_layers):
        p = f"model.layers.{i}."
        w = sd[p + "attention.wqkv.weight"].view(hkv, g + 2, d, -1)
        out[f"L{i}.qkv"] = torch.cat([w[:, :g].reshape(hq * d, -1), w[:, g].reshape(hkv * d, -1), w[:, g + 1].reshape(hkv * d, -1)], 0)
        if p + "attention.wqkv.bias" in sd:
            b = sd[p + "attention.wqkv.bias"].view(hkv, g + 2, d)
            out[f"L{i}.qkv.bias"] = torch.cat([b[:, :g].reshape(-1), b[:, g].reshape(-1), b[:, g + 1].reshape(-1)], 0)
        out[f"L{i}.o"] = sd[p + "attention.wo.weight"]
        out[f"L{i}.input_norm"] = sd[p + "attention_norm.weight"]
        out[f"L{i}.post_norm"] = sd[p + "ffn_norm.weight"]
        out[f"L{i}.gate"] = sd[p + "feed_forward.w1.weight"]
        out[f"L{i}.up"] = sd[p + "feed_forward.w3.weight"]
        out[f"L{i}.down"] = sd[p + "feed_forward.w2.weight"]
    return dc, out


def convert_minicpm(cfg: dict, sd: StateDict):
    """MiniCPM 1B/2B (reference models/minicpm.py:58-66 ``apply_residual_scale``): llama with an embedding scale, a
    residual scale ``scale_depth / sqrt(layers)`` (folded into o_proj / down_proj) and logits divided by
    ``hidden / dim_model_base``."""
    dc = _base_config(cfg, "minicpm", tie_word_embeddings=True)
    dc.embed_scale = float(_get(cfg, "scale_emb", default=1.0))
    dc.logit_scale = float(_get(cfg, "dim_model_base", default=dc.hidden_size)) / dc.hidden_size
    rs = float(_get(cfg, "scale_depth", default=1.0)) / math.sqrt(dc.num_layers)
    return dc, _llama_keys(cfg, sd, dc, o_scale=rs, down_scale=rs)


def _mla_layout(cfg: dict):
    """(nope, rope, v, padded head_dim) of a latent-attention config"""
    n, r, vd = cfg["qk_nope_head_dim"], cfg["qk_rope_head_dim"], cfg["v_head_dim"]
    need = max(n + r, vd)
    fits = [x for x in ENGINE_HEAD_DIMS if x >= need]
    if not fits:
        raise NotImplementedError(f"latent attention with {need} channels per head")
    return n, r, vd, min(fits)


def _mla_keys(cfg: dict, sd: StateDict, out: StateDict, i: int, o_scale: float = 1.0, pair_up: bool = False):
    """Latent-attention projections of layer ``i`` in the engine's head layout (decoder.py::MLAWeights): q heads
    [rotary | no-position | 0], K heads [0 (shared rotary key goes here) | no-position | 0], V heads [v | 0]."""
    n, r, vd, d = _mla_layout(cfg)
    heads = cfg["num_attention_heads"]
    p = f"model.layers.{i}.self_attn."
    # pair_up: the checkpoint rotates channel pairs (j, j + r/2); the rotary channels are re-ordered so that the pairs
    # sit side by side (2j, 2j + 1) in q and in the shared key alike (scores do not change), which is the order the
    # attention kernel rotates and appends in by itself when only part of the head is rotary
    order = torch.arange(r)
    if pair_up:
        order = torch.stack([order[: r // 2], order[r // 2:]], 1).reshape(-1)

    def q_rows(w):                      # [heads * (n + r), in] -> [heads * d, in]
        w = w.reshape(heads, n + r, -1)
        z = torch.zeros(heads, d - n - r, w.shape[2], dtype=w.dtype)
        return torch.cat([w[:, n:][:, order], w[:, :n], z], 1).reshape(heads * d, -1)

    kv_a = sd[p + "kv_a_proj_with_mqa.weight"]
    kv_a = torch.cat([kv_a[:-r], kv_a[-r:][order]], 0)
    if cfg.get("q_lora_rank"):
        # both latent projections read the same rows: one stacked weight [q latent ; kv latent | rotary key]
        out[f"L{i}.kv_a"] = torch.cat([sd[p + "q_a_proj.weight"], kv_a], 0)
        out[f"L{i}.q_a_norm"] = sd[p + "q_a_layernorm.weight"]
        out[f"L{i}.q_b"] = q_rows(sd[p + "q_b_proj.weight"])
    else:
        out[f"L{i}.q"] = q_rows(sd[p + "q_proj.weight"])
        out[f"L{i}.kv_a"] = kv_a
    out[f"L{i}.kv_a_norm"] = sd[p + "kv_a_layernorm.weight"]
    kvb = sd[p + "kv_b_proj.weight"].reshape(heads, n + vd, -1)
    lora = kvb.shape[2]
    k = torch.cat([torch.zeros(heads, r, lora, dtype=kvb.dtype), kvb[:, :n],
                   torch.zeros(heads, d - r - n, lora, dtype=kvb.dtype)], 1)
    v = torch.cat([kvb[:, n:], torch.zeros(heads, d - vd, lora, dtype=kvb.dtype)], 1)
    out[f"L{i}.kv_b"] = torch.cat([k.reshape(heads * d, lora), v.reshape(heads * d, lora)], 0)
    o = sd[p + "o_proj.weight"].reshape(-1, heads, vd) * o_scale
    out[f"L{i}.o"] = torch.cat([o, torch.zeros(o.shape[0], heads, d - vd, dtype=o.dtype)], 2).reshape(o.shape[0], heads * d)


def convert_minicpm3(cfg: dict, sd: StateDict):
    """MiniCPM3-4B (reference models/minicpm3.py): MiniCPM's embedding / residual / logit scales plus multi-head
    latent attention; rotate-half RoPE on the ``qk_rope_head_dim`` rotary channels (stored pair by pair here), softmax scale
    1 / sqrt(qk_nope + qk_rope)."""
    dc = _probe_minicpm3(cfg)
    rs = float(_get(cfg, "scale_depth", default=1.0)) / math.sqrt(dc.num_layers)
    out = {"embed": sd["model.embed_tokens.weight"], "final_norm": sd["model.norm.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    for i in range(dc.num_layers):
        p = f"model.layers.{i}."
        _mla_keys(cfg, sd, out, i, o_scale=rs, pair_up=True)
        out[f"L{i}.input_norm"] = sd[p + "input_layernorm.weight"]
        out[f"L{i}.post_norm"] = sd[p + "post_attention_layernorm.weight"]
        out[f"L{i}.gate"] = sd[p + "mlp.gate_proj.weight"]
        out[f"L{i}.up"] = sd[p + "mlp.up_proj.weight"]
        out[f"L{i}.down"] = sd[p + "mlp.down_proj.weight"] * rs
    return dc, out


def _probe_deepseek_v2(cfg):
    n, r, vd, d = _mla_layout(cfg)
    heads = cfg["num_attention_heads"]
    grouped = _get(cfg, "topk_method", default="greedy") == "group_limited_greedy" and (_get(cfg, "n_group", default=1) or 1) > 1
    scale = 1.0 / math.sqrt(n + r)
    rs = _get(cfg, "rope_scaling") or cfg.get("rope_parameters") or {}
    if rs.get("mscale_all_dim") and rs.get("factor", 1.0) > 1.0:      # yarn: the softmax scale carries mscale^2
        ms = 0.1 * float(rs["mscale_all_dim"]) * math.log(float(rs["factor"])) + 1.0
        scale *= ms * ms
    return _base_config(cfg, "deepseek_v2", num_kv_heads=heads, head_dim=d, rot_dim=r, rope_interleaved=True,
                        attn_scale=scale, mla_q_lora=int(cfg.get("q_lora_rank") or 0), mla_kv_lora=int(cfg["kv_lora_rank"]),
                        num_experts=int(cfg["n_routed_experts"]), top_k=int(cfg["num_experts_per_tok"]),
                        moe_intermediate_size=int(cfg["moe_intermediate_size"]), norm_topk_prob=False,
                        moe_n_group=int(cfg["n_group"]) if grouped else 1,
                        moe_topk_group=int(cfg["topk_group"]) if grouped else 1,
                        shared_expert_intermediate_size=int(cfg["moe_intermediate_size"]) * int(cfg.get("n_shared_experts") or 0))


def convert_deepseek_v2(cfg: dict, sd: StateDict):
    """DeepSeek-V2 / V2-Lite (the reference converts the linears only and runs the model's own remote-code forward,
    example/GPU/HuggingFace/LLM/deepseek):
    latent attention with interleaved rotary pairs, dense MLP in the first ``first_k_dense_replace`` layers, then
    routed experts (softmax scores, plain or group-limited top-k without renormalization, ``routed_scaling_factor`` folded into the experts'
    down projections) plus always-on shared experts.  Expert weights are taken stacked (``experts.gate_up_proj``) or
    per expert (``experts.N.gate_proj``)."""
    dc = _probe_deepseek_v2(cfg)
    e, inter = dc.num_experts, dc.moe_intermediate_size
    rsf = float(_get(cfg, "routed_scaling_factor", default=1.0))
    out = {"embed": sd["model.embed_tokens.weight"], "final_norm": sd["model.norm.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    for i in range(dc.num_layers):
        p = f"model.layers.{i}."
        _mla_keys(cfg, sd, out, i)
        out[f"L{i}.input_norm"] = sd[p + "input_layernorm.weight"]
        out[f"L{i}.post_norm"] = sd[p + "post_attention_layernorm.weight"]
        if p + "mlp.gate.weight" not in sd:
            for n in ("gate", "up", "down"):
                out[f"L{i}.{n}"] = sd[p + f"mlp.{n}_proj.weight"]
            continue
        out[f"L{i}.router"] = sd[p + "mlp.gate.weight"]
        if p + "mlp.experts.gate_up_proj" in sd:
            gu = sd[p + "mlp.experts.gate_up_proj"]                     # [E, 2 I, H]
            gate, up, down = gu[:, :inter], gu[:, inter:], sd[p + "mlp.experts.down_proj"]
        else:
            gate, up, down = (torch.stack([sd[p + f"mlp.experts.{j}.{n}_proj.weight"] for j in range(e)])
                              for n in ("gate", "up", "down"))
        out[f"L{i}.experts.gate"] = gate.reshape(e * inter, -1)
        out[f"L{i}.experts.up"] = up.reshape(e * inter, -1)
        out[f"L{i}.experts.down"] = (down * rsf).reshape(e * down.shape[1], -1)
        if dc.shared_expert_intermediate_size:
            for n in ("gate", "up", "down"):
                out[f"L{i}.shared.{n}"] = sd[p + f"mlp.shared_experts.{n}_proj.weight"]
    return dc, out


def _probe_deepseek_v3(cfg):
    dc = _probe_deepseek_v2(cfg)
    dc.model_type = "deepseek_v3"
    dc.moe_score, dc.moe_group_score = "sigmoid", "top2sum"
    dc.norm_topk_prob = bool(_get(cfg, "norm_topk_prob", default=True))
    dc.moe_n_group, dc.moe_topk_group = int(_get(cfg, "n_group", default=1) or 1), int(_get(cfg, "topk_group", default=1) or 1)
    dc.rope_interleaved = bool(_get(cfg, "rope_interleave", default=True))
    return dc


def convert_deepseek_v3(cfg: dict, sd: StateDict):
    """DeepSeek-V3 / R1 (reference: example/GPU/CPU-GPU-Hybrid-DeepSeek-R1-Inference): the V2 layout with sigmoid router scores, a learned selection bias
    (``e_score_correction_bias``), groups ranked by the sum of their two best experts, renormalized weights."""
    dc, out = convert_deepseek_v2(cfg, sd)
    v3 = _probe_deepseek_v3(cfg)
    for i in range(dc.num_layers):
        b = sd.get(f"model.layers.{i}.mlp.gate.e_score_correction_bias")
        if b is not None:
            out[f"L{i}.router_bias"] = b
    return v3, out


def convert_llama_like(cfg: dict, sd: StateDict):
    """Aquila / Aquila2, Yi and other checkpoints that use llama's tensor names under another ``model_type``
    (reference models/aquila.py)."""
    dc = _base_config(cfg, cfg.get("model_type", "llama"))
    return dc, _llama_keys(cfg, sd, dc)


def _cfg_decilm(cfg: dict) -> DecoderConfig:
    per = [int(n) for n in cfg["num_key_value_heads_per_layer"]]
    common = max(per)
    if any(common % n for n in per):
        raise NotImplementedError(f"kv head counts {sorted(set(per))} have no common multiple among them")
    return _base_config(cfg, "deci", num_kv_heads=common)


def convert_decilm(cfg: dict, sd: StateDict):
    """DeciLM-7B (``model_type: deci``; reference models/decilm.py:48): llama layers whose number of key / value heads
    differs from layer to layer (``num_key_value_heads_per_layer``).  The engine has one KV layout for all layers, so
    every layer is brought to the largest count by repeating its k / v projection rows head by head: query head j of a
    layer with n kv heads reads kv head j // (heads / n), and after each kv head is repeated r = max / n times in place
    it reads head j // (heads / max) of the repeated set, which is the same original head.  Exact; the price is KV
    cache memory for the repeated heads."""
    dc = _cfg_decilm(cfg)
    d, common = dc.head_dim, dc.num_kv_heads
    wide = dict(sd)
    for i, n in enumerate(cfg["num_key_value_heads_per_layer"]):
        r = common // int(n)
        if r > 1:
            for name in ("k_proj", "v_proj"):
                key = f"model.layers.{i}.self_attn.{name}.weight"
                w = sd[key]
                wide[key] = w.view(int(n), d, w.shape[1]).repeat_interleave(r, 0).reshape(common * d, w.shape[1])
    return dc, _llama_keys(cfg, wide, dc)


# ------------------------------------------------------------------------------------------------
# LayerNorm families (HuggingFace has classes for them; the native engine is opt-in: ``native_engine=True`` or
# BIGDL_NATIVE_ENGINE=1).  Reference forwards: models/gptneox.py, bloom.py, mpt.py, falcon.py, gpt2.py,
# gptbigcode.py, phi.py, starcoder2.py, stablelm.py (attention rebinding per family; here one engine with flags).
# ------------------------------------------------------------------------------------------------
def _ln_config(cfg: dict, model_type: str, **over) -> DecoderConfig:
    kw = dict(norm="layer", gated_mlp=False,
              rms_eps=_get(cfg, "layer_norm_eps", "layer_norm_epsilon", "norm_eps", "norm_epsilon", default=1e-5))
    kw.update(over)
    return _base_config(cfg, model_type, **kw)


def _deinterleave_qkv(w, heads, d):
    """[heads, (q, k, v), d, ...] rows -> q | k | v"""
    return w.view(heads, 3, d, *w.shape[1:]).transpose(0, 1).reshape(3 * heads * d, *w.shape[1:])


def _grouped_qkv(w, hq, hkv, d):
    """per kv head: its group of query heads, then k, then v  ->  q | k | v"""
    g = hq // hkv
    w = w.view(hkv, g + 2, d, *w.shape[1:])
    tail = w.shape[3:]
    return torch.cat([w[:, :g].reshape(hq * d, *tail), w[:, g].reshape(hkv * d, *tail), w[:, g + 1].reshape(hkv * d, *tail)], 0)


def _put(out, key, sd, name, fn=None):
    """copy ``name`` (+ ``.bias``) of the checkpoint to canonical ``key``"""
    for suffix, ck in ((".weight", key), (".bias", key + ".bias")):
        if name + suffix in sd:
            t = sd[name + suffix]
            out[ck] = fn(t) if fn is not None else t


def _cfg_gpt_neox(cfg):
    d = cfg["hidden_size"] // cfg["num_attention_heads"]
    rp = cfg.get("rope_parameters") or {}
    pct = _get(cfg, "rotary_pct", "partial_rotary_factor", default=rp.get("partial_rotary_factor", 0.25))
    return _ln_config(cfg, "gpt_neox", act=_get(cfg, "hidden_act", default="gelu"),
                      parallel_residual=bool(_get(cfg, "use_parallel_residual", default=True)),
                      rot_dim=int(d * pct) if pct != 1.0 else None,
                      rope_theta=float(_get(cfg, "rotary_emb_base", "rope_theta", default=rp.get("rope_theta", 10000.0))))


def convert_gpt_neox(cfg: dict, sd: StateDict):
    """GPT-NeoX / Pythia / RedPajama / dolly-v2 (reference models/gptneox.py:35-151): per-head interleaved
    ``query_key_value``, partial rotate-half rotary, parallel residual with two LayerNorms."""
    dc = _cfg_gpt_neox(cfg)
    out = {"embed": sd["gpt_neox.embed_in.weight"]}
    out["lm_head"] = sd.get("embed_out.weight", sd.get("lm_head.weight", out["embed"]))
    _put(out, "final_norm", sd, "gpt_neox.final_layer_norm")
    for i in range(dc.num_layers):
        p = f"gpt_neox.layers.{i}."
        _put(out, f"L{i}.qkv", sd, p + "attention.query_key_value", lambda t: _deinterleave_qkv(t, dc.num_heads, dc.head_dim))
        _put(out, f"L{i}.o", sd, p + "attention.dense")
        _put(out, f"L{i}.input_norm", sd, p + "input_layernorm")
        _put(out, f"L{i}.post_norm", sd, p + "post_attention_layernorm")
        _put(out, f"L{i}.up", sd, p + "mlp.dense_h_to_4h")
        _put(out, f"L{i}.down", sd, p + "mlp.dense_4h_to_h")
    return dc, out


def _cfg_bloom(cfg):
    return _ln_config(cfg, "bloom", act="gelu_tanh", alibi=True, intermediate_size=4 * _get(cfg, "hidden_size", "n_embed"),
                      hidden_size=_get(cfg, "hidden_size", "n_embed"), tie_word_embeddings=True)


def convert_bloom(cfg: dict, sd: StateDict):
    """BLOOM (reference models/bloom.py:60-208): ALiBi, LayerNorm on the embedding, interleaved qkv, tanh-GELU."""
    if _get(cfg, "apply_residual_connection_post_layernorm", default=False):
        raise NotImplementedError("bloom with apply_residual_connection_post_layernorm")
    dc = _cfg_bloom(cfg)
    pre = "transformer." if "transformer.word_embeddings.weight" in sd else ""
    out = {"embed": sd[pre + "word_embeddings.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    _put(out, "embed_norm", sd, pre + "word_embeddings_layernorm")
    _put(out, "final_norm", sd, pre + "ln_f")
    for i in range(dc.num_layers):
        p = f"{pre}h.{i}."
        _put(out, f"L{i}.qkv", sd, p + "self_attention.query_key_value", lambda t: _deinterleave_qkv(t, dc.num_heads, dc.head_dim))
        _put(out, f"L{i}.o", sd, p + "self_attention.dense")
        _put(out, f"L{i}.input_norm", sd, p + "input_layernorm")
        _put(out, f"L{i}.post_norm", sd, p + "post_attention_layernorm")
        _put(out, f"L{i}.up", sd, p + "mlp.dense_h_to_4h")
        _put(out, f"L{i}.down", sd, p + "mlp.dense_4h_to_h")
    return dc, out


def _mpt_slopes(n_heads: int, bias_max: float = 8.0):
    n2 = 2 ** math.ceil(math.log2(n_heads))
    sl = [1.0 / 2.0 ** ((i + 1) * bias_max / n2) for i in range(n2)]
    if n2 != n_heads:
        sl = (sl[1::2] + sl[::2])[:n_heads]
    return tuple(sl)


def _cfg_mpt(cfg):
    att = cfg.get("attn_config") or {}
    if att.get("qk_ln") or att.get("clip_qkv") or not att.get("alibi", True):
        raise NotImplementedError("mpt with qk_ln / clip_qkv / learned positions")
    heads, hidden = cfg["n_heads"], cfg["d_model"]
    return _ln_config(cfg, "mpt", act="gelu", alibi=True, alibi_slopes=_mpt_slopes(heads, att.get("alibi_bias_max", 8)),
                      hidden_size=hidden, num_heads=heads, num_kv_heads=heads, head_dim=hidden // heads,
                      num_layers=cfg["n_layers"], intermediate_size=int(cfg.get("expansion_ratio", 4) * hidden),
                      max_position=cfg.get("max_seq_len", 2048), tie_word_embeddings=True,
                      attn_scale=att.get("softmax_scale"), rms_eps=float(cfg.get("layer_norm_epsilon", 1e-5)))


def convert_mpt(cfg: dict, sd: StateDict):
    """MPT (reference models/mpt.py:29-151): ALiBi with mpt's slope ordering, bias-free LayerNorm, ``Wqkv`` = q | k | v."""
    dc = _cfg_mpt(cfg)
    out = {"embed": sd["transformer.wte.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    _put(out, "final_norm", sd, "transformer.norm_f")
    for i in range(dc.num_layers):
        p = f"transformer.blocks.{i}."
        _put(out, f"L{i}.qkv", sd, p + "attn.Wqkv")
        _put(out, f"L{i}.o", sd, p + "attn.out_proj")
        _put(out, f"L{i}.input_norm", sd, p + "norm_1")
        _put(out, f"L{i}.post_norm", sd, p + "norm_2")
        _put(out, f"L{i}.up", sd, p + "ffn.up_proj")
        _put(out, f"L{i}.down", sd, p + "ffn.down_proj")
    return dc, out


def _cfg_falcon(cfg):
    heads = _get(cfg, "num_attention_heads", "n_head")
    hidden = cfg["hidden_size"]
    d = hidden // heads
    new_arch = bool(cfg.get("new_decoder_architecture", False))
    if new_arch:
        nkv = _get(cfg, "num_kv_heads", default=heads)
    else:
        nkv = 1 if cfg.get("multi_query", True) else heads
    alibi = bool(cfg.get("alibi", False))
    over = {}
    if alibi:      # falcon scales the ALiBi bias together with the scores: slope / sqrt(d)
        over = dict(alibi=True, alibi_slopes=tuple((alibi_slopes(heads) / math.sqrt(d)).tolist()))
    return _ln_config(cfg, "falcon", act=_get(cfg, "activation", default="gelu"), num_heads=heads, num_kv_heads=nkv, head_dim=d,
                      num_layers=_get(cfg, "num_hidden_layers", "n_layer"),
                      intermediate_size=_get(cfg, "ffn_hidden_size", default=4 * hidden),
                      parallel_residual=new_arch or bool(cfg.get("parallel_attn", True)),
                      max_position=_get(cfg, "max_position_embeddings", default=2048), **over)


def convert_falcon(cfg: dict, sd: StateDict):
    """Falcon 7B / 40B / 180B / RW (reference models/falcon.py): multi-query (q | k | v), grouped (new decoder
    architecture: per kv group) or per-head interleaved (RW) fused qkv; parallel attention + MLP."""
    dc = _cfg_falcon(cfg)
    new_arch = bool(cfg.get("new_decoder_architecture", False))
    out = {"embed": sd["transformer.word_embeddings.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    _put(out, "final_norm", sd, "transformer.ln_f")
    hq, hkv, d = dc.num_heads, dc.num_kv_heads, dc.head_dim
    for i in range(dc.num_layers):
        p = f"transformer.h.{i}."
        if new_arch:
            fn = lambda t: _grouped_qkv(t, hq, hkv, d)      # noqa: E731
        elif hkv == 1:
            fn = None
        else:
            fn = lambda t: _deinterleave_qkv(t, hq, d)      # noqa: E731
        _put(out, f"L{i}.qkv", sd, p + "self_attention.query_key_value", fn)
        _put(out, f"L{i}.o", sd, p + "self_attention.dense")
        if p + "ln_attn.weight" in sd:
            _put(out, f"L{i}.input_norm", sd, p + "ln_attn")
            _put(out, f"L{i}.post_norm", sd, p + "ln_mlp")
        else:
            _put(out, f"L{i}.input_norm", sd, p + "input_layernorm")
            if not dc.parallel_residual:
                _put(out, f"L{i}.post_norm", sd, p + "post_attention_layernorm")
        _put(out, f"L{i}.up", sd, p + "mlp.dense_h_to_4h")
        _put(out, f"L{i}.down", sd, p + "mlp.dense_4h_to_h")
    return dc, out


def _cfg_gpt2(cfg):
    hidden, heads = _get(cfg, "n_embd", "hidden_size"), _get(cfg, "n_head", "num_attention_heads")
    if cfg.get("scale_attn_by_inverse_layer_idx") or cfg.get("reorder_and_upcast_attn"):
        raise NotImplementedError("gpt2 with per-layer attention scaling")
    return _ln_config(cfg, "gpt2", act=_get(cfg, "activation_function", default="gelu_new"), no_rope=True,
                      hidden_size=hidden, num_heads=heads, num_kv_heads=heads, head_dim=hidden // heads,
                      intermediate_size=_get(cfg, "n_inner", default=4 * hidden), max_position=_get(cfg, "n_positions", default=1024),
                      tie_word_embeddings=True)


def convert_gpt2(cfg: dict, sd: StateDict):
    """GPT-2 (reference models/gpt2.py): learned positions, ``Conv1D`` weights stored [in, out]."""
    dc = _cfg_gpt2(cfg)
    pre = "transformer." if "transformer.wte.weight" in sd else ""
    out = {"embed": sd[pre + "wte.weight"], "pos_embed": sd[pre + "wpe.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    _put(out, "final_norm", sd, pre + "ln_f")
    tr = lambda t: t.t().contiguous() if t.dim() == 2 else t      # noqa: E731
    for i in range(dc.num_layers):
        p = f"{pre}h.{i}."
        _put(out, f"L{i}.qkv", sd, p + "attn.c_attn", tr)
        _put(out, f"L{i}.o", sd, p + "attn.c_proj", tr)
        _put(out, f"L{i}.input_norm", sd, p + "ln_1")
        _put(out, f"L{i}.post_norm", sd, p + "ln_2")
        _put(out, f"L{i}.up", sd, p + "mlp.c_fc", tr)
        _put(out, f"L{i}.down", sd, p + "mlp.c_proj", tr)
    return dc, out


def _cfg_gpt_bigcode(cfg):
    hidden, heads = cfg["n_embd"], cfg["n_head"]
    return _ln_config(cfg, "gpt_bigcode", act=_get(cfg, "activation_function", default="gelu_pytorch_tanh"), no_rope=True,
                      hidden_size=hidden, num_heads=heads, num_kv_heads=1 if cfg.get("multi_query", True) else heads,
                      head_dim=hidden // heads, intermediate_size=_get(cfg, "n_inner", default=4 * hidden),
                      max_position=_get(cfg, "n_positions", default=8192), tie_word_embeddings=True)


def convert_gpt_bigcode(cfg: dict, sd: StateDict):
    """StarCoder / SantaCoder (reference models/gptbigcode.py): learned positions, multi-query ``c_attn`` = q | k | v."""
    dc = _cfg_gpt_bigcode(cfg)
    out = {"embed": sd["transformer.wte.weight"], "pos_embed": sd["transformer.wpe.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    _put(out, "final_norm", sd, "transformer.ln_f")
    fn = None if dc.num_kv_heads == 1 else (lambda t: _deinterleave_qkv(t, dc.num_heads, dc.head_dim))
    for i in range(dc.num_layers):
        p = f"transformer.h.{i}."
        _put(out, f"L{i}.qkv", sd, p + "attn.c_attn", fn)
        _put(out, f"L{i}.o", sd, p + "attn.c_proj")
        _put(out, f"L{i}.input_norm", sd, p + "ln_1")
        _put(out, f"L{i}.post_norm", sd, p + "ln_2")
        _put(out, f"L{i}.up", sd, p + "mlp.c_fc")
        _put(out, f"L{i}.down", sd, p + "mlp.c_proj")
    return dc, out


def _cfg_opt(cfg):
    if cfg.get("word_embed_proj_dim", cfg["hidden_size"]) != cfg["hidden_size"] or not cfg.get("do_layer_norm_before", True):
        raise NotImplementedError("opt with projected embeddings / post-LayerNorm (opt-350m)")
    return _ln_config(cfg, "opt", act=_get(cfg, "activation_function", default="relu"), no_rope=True, pos_offset=2,
                      intermediate_size=cfg["ffn_dim"], tie_word_embeddings=True)


def convert_opt(cfg: dict, sd: StateDict):
    """OPT: learned positions with an index offset of 2, ReLU MLP, separate q / k / v with bias."""
    dc = _cfg_opt(cfg)
    pre = "model.decoder."
    out = {"embed": sd[pre + "embed_tokens.weight"], "pos_embed": sd[pre + "embed_positions.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    _put(out, "final_norm", sd, pre + "final_layer_norm")
    for i in range(dc.num_layers):
        p = f"{pre}layers.{i}."
        out[f"L{i}.qkv"] = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        out[f"L{i}.qkv.bias"] = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        _put(out, f"L{i}.o", sd, p + "self_attn.out_proj")
        _put(out, f"L{i}.input_norm", sd, p + "self_attn_layer_norm")
        _put(out, f"L{i}.post_norm", sd, p + "final_layer_norm")
        _put(out, f"L{i}.up", sd, p + "fc1")
        _put(out, f"L{i}.down", sd, p + "fc2")
    return dc, out


def _partial_rot(cfg, d, default=1.0):
    rp = cfg.get("rope_parameters") or {}
    pct = _get(cfg, "partial_rotary_factor", "rope_pct", default=rp.get("partial_rotary_factor", default))
    theta = float(_get(cfg, "rope_theta", default=rp.get("rope_theta", 10000.0)))
    return (int(d * pct) if pct != 1.0 else None), theta


def _sep_qkv(out, sd, p, i, names=("q_proj", "k_proj", "v_proj")):
    out[f"L{i}.qkv"] = torch.cat([sd[p + n + ".weight"] for n in names], 0)
    if p + names[0] + ".bias" in sd:
        out[f"L{i}.qkv.bias"] = torch.cat([sd[p + n + ".bias"] for n in names], 0)


def _cfg_phi(cfg):
    if cfg.get("qk_layernorm"):
        raise NotImplementedError("phi with qk_layernorm")
    d = cfg["hidden_size"] // cfg["num_attention_heads"]
    rot, theta = _partial_rot(cfg, d, 0.5)
    return _ln_config(cfg, "phi", act=_get(cfg, "hidden_act", default="gelu_new"), parallel_residual=True, rot_dim=rot,
                      rope_theta=theta)


def convert_phi(cfg: dict, sd: StateDict):
    """Phi-1 / 1.5 / 2 (reference models/phi.py:43-162): one LayerNorm feeds attention and MLP in parallel, partial
    rotary, biases everywhere including the lm_head."""
    dc = _cfg_phi(cfg)
    out = {"embed": sd["model.embed_tokens.weight"]}
    _put(out, "lm_head", sd, "lm_head")
    _put(out, "final_norm", sd, "model.final_layernorm")
    for i in range(dc.num_layers):
        p = f"model.layers.{i}."
        _sep_qkv(out, sd, p + "self_attn.", i)
        _put(out, f"L{i}.o", sd, p + "self_attn.dense")
        _put(out, f"L{i}.input_norm", sd, p + "input_layernorm")
        _put(out, f"L{i}.up", sd, p + "mlp.fc1")
        _put(out, f"L{i}.down", sd, p + "mlp.fc2")
    return dc, out


def _cfg_starcoder2(cfg):
    rot, theta = _partial_rot(cfg, cfg["hidden_size"] // cfg["num_attention_heads"])
    return _ln_config(cfg, "starcoder2", act=_get(cfg, "hidden_act", default="gelu_pytorch_tanh"), rot_dim=rot, rope_theta=theta,
                      rms_eps=_get(cfg, "norm_epsilon", default=1e-5), sliding_window=int(cfg.get("sliding_window") or 0),
                      tie_word_embeddings=bool(cfg.get("tie_word_embeddings", True)))


def convert_starcoder2(cfg: dict, sd: StateDict):
    """StarCoder2 (reference models/starcoder2.py:56-140): llama-shaped blocks with LayerNorm, biases, a plain
    ``c_fc`` / ``c_proj`` MLP, grouped-query attention and a sliding window."""
    dc = _cfg_starcoder2(cfg)
    out = {"embed": sd["model.embed_tokens.weight"]}
    out["lm_head"] = sd.get("lm_head.weight", out["embed"])
    _put(out, "final_norm", sd, "model.norm")
    for i in range(dc.num_layers):
        p = f"model.layers.{i}."
        _sep_qkv(out, sd, p + "self_attn.", i)
        _put(out, f"L{i}.o", sd, p + "self_attn.o_proj")
        _put(out, f"L{i}.input_norm", sd, p + "input_layernorm")
        _put(out, f"L{i}.post_norm", sd, p + "post_attention_layernorm")
        _put(out, f"L{i}.up", sd, p + "mlp.c_fc")
        _put(out, f"L{i}.down", sd, p + "mlp.c_proj")
    return dc, out


def _cfg_stablelm(cfg):
    if cfg.get("qk_layernorm"):
        raise NotImplementedError("stablelm with per-head qk LayerNorm")
    rot, theta = _partial_rot(cfg, cfg["hidden_size"] // cfg["num_attention_heads"], 0.25)
    return _ln_config(cfg, "stablelm", act=_get(cfg, "hidden_act", default="silu"), gated_mlp=True, rot_dim=rot, rope_theta=theta,
                      parallel_residual=bool(cfg.get("use_parallel_residual", False)))


def convert_stablelm(cfg: dict, sd: StateDict):
    """StableLM (reference models/stablelm.py:62-200): llama's gated MLP and module names with LayerNorm and
    25 % rotary."""
    dc = _cfg_stablelm(cfg)
    out = {"embed": sd["model.embed_tokens.weight"], "lm_head": sd["lm_head.weight"]}
    _put(out, "final_norm", sd, "model.norm")
    for i in range(dc.num_layers):
        p = f"model.layers.{i}."
        _sep_qkv(out, sd, p + "self_attn.", i)
        _put(out, f"L{i}.o", sd, p + "self_attn.o_proj")
        _put(out, f"L{i}.input_norm", sd, p + "input_layernorm")
        if not dc.parallel_residual:
            _put(out, f"L{i}.post_norm", sd, p + "post_attention_layernorm")
        for n in ("gate", "up", "down"):
            _put(out, f"L{i}.{n}", sd, p + f"mlp.{n}_proj")
    return dc, out


def _cfg_gptj(cfg):
    hidden, heads = cfg["n_embd"], cfg["n_head"]
    d = hidden // heads
    rot = cfg.get("rotary_dim") or d
    return _ln_config(cfg, "gptj", act=_get(cfg, "activation_function", default="gelu_new"), parallel_residual=True,
                      hidden_size=hidden, num_heads=heads, num_kv_heads=heads, head_dim=d, rot_dim=rot if rot != d else None,
                      rope_interleaved=True, intermediate_size=_get(cfg, "n_inner", default=4 * hidden),
                      max_position=_get(cfg, "n_positions", default=2048))


def convert_gptj(cfg: dict, sd: StateDict):
    """GPT-J (reference models/gptj.py): interleaved partial rotary, one LayerNorm for the parallel attention + MLP."""
    dc = _cfg_gptj(cfg)
    out = {"embed": sd["transformer.wte.weight"]}
    _put(out, "lm_head", sd, "lm_head")
    _put(out, "final_norm", sd, "transformer.ln_f")
    for i in range(dc.num_layers):
        p = f"transformer.h.{i}."
        _sep_qkv(out, sd, p + "attn.", i)
        _put(out, f"L{i}.o", sd, p + "attn.out_proj")
        _put(out, f"L{i}.input_norm", sd, p + "ln_1")
        _put(out, f"L{i}.up", sd, p + "mlp.fc_in")
        _put(out, f"L{i}.down", sd, p + "mlp.fc_out")
    return dc, out


ENGINE_HEAD_DIMS = (64, 96, 128, 256)


def pad_head_dim(dc: DecoderConfig, sd: StateDict):
    """Bring an odd head_dim (80: phi-2 / stablelm-3b, 72, 112, 160, 192) to the next size the attention kernels are
    built for by zero-padding every head: extra q / k channels add 0 to the scores, extra v channels produce zeros
    that the zero columns of o_proj ignore.  The softmax scale and the rotary dimension keep their original values
    (reference: models/common.py:55-115 padding helpers, models/minicpm3.py ``padding_v_head_dim``)."""
    d = dc.head_dim
    if d in ENGINE_HEAD_DIMS or d > max(ENGINE_HEAD_DIMS):
        return dc, sd
    import dataclasses
    nd = min(x for x in ENGINE_HEAD_DIMS if x > d)
    hq, hkv = dc.num_heads, dc.num_kv_heads
    out = dict(sd)

    def pad_rows(t, heads):          # [heads * d, ...] -> [heads * nd, ...]
        t = t.reshape(heads, d, *t.shape[1:])
        z = torch.zeros(heads, nd - d, *t.shape[2:], dtype=t.dtype)
        return torch.cat([t, z], 1).reshape(heads * nd, *t.shape[2:])

    for i in range(dc.num_layers if sd else 0):
        for key in (f"L{i}.qkv", f"L{i}.qkv.bias"):
            if key in sd:
                t = sd[key]
                q, k, v = t[: hq * d], t[hq * d:(hq + hkv) * d], t[(hq + hkv) * d:]
                out[key] = torch.cat([pad_rows(q, hq), pad_rows(k, hkv), pad_rows(v, hkv)], 0)
        o = sd[f"L{i}.o"]                                              # [hidden, hq * d]
        o = o.reshape(o.shape[0], hq, d)
        out[f"L{i}.o"] = torch.cat([o, torch.zeros(o.shape[0], hq, nd - d, dtype=o.dtype)], 2).reshape(o.shape[0], hq * nd)
    ndc = dataclasses.replace(dc, head_dim=nd, rot_dim=dc.rot_dim or d,
                              attn_scale=dc.attn_scale if dc.attn_scale is not None else 1.0 / math.sqrt(d))
    return ndc, out


LN_CONVERTERS = {"gpt_neox": convert_gpt_neox, "bloom": convert_bloom, "mpt": convert_mpt, "falcon": convert_falcon,
                 "gpt2": convert_gpt2, "gpt_bigcode": convert_gpt_bigcode, "opt": convert_opt, "phi": convert_phi,
                 "starcoder2": convert_starcoder2, "stablelm": convert_stablelm, "gptj": convert_gptj}
LN_CONFIGS = {"gpt_neox": _cfg_gpt_neox, "bloom": _cfg_bloom, "mpt": _cfg_mpt, "falcon": _cfg_falcon, "gpt2": _cfg_gpt2,
              "gpt_bigcode": _cfg_gpt_bigcode, "opt": _cfg_opt, "phi": _cfg_phi, "starcoder2": _cfg_starcoder2,
              "stablelm": _cfg_stablelm, "gptj": _cfg_gptj}


ARCH_CONVERTERS: Dict[str, Callable[[dict, StateDict], Tuple[DecoderConfig, StateDict]]] = {
    "chatglm": convert_chatglm, "baichuan": convert_baichuan, "qwen": convert_qwen, "internlm": convert_internlm,
    "internlm2": convert_internlm2, "minicpm": convert_minicpm, "minicpm3": convert_minicpm3, "deepseek_v2": convert_deepseek_v2, "deepseek_v3": convert_deepseek_v3, "aquila": convert_llama_like, "yi": convert_llama_like,
    "deci": convert_decilm,
    **LN_CONVERTERS,
}


# ------------------------------------------------------------------------------------------------
# plain PyTorch forward over the canonical form (CPU path and test oracle)
# ------------------------------------------------------------------------------------------------
def _rope(x, pos, inv_freq, rot, interleaved, scaling):
    """x [B, H, S, D]; rotary on the first ``rot`` features (half-split or interleaved pairs)."""
    ang = pos[:, None].float() * inv_freq[None].float()                 # [S, rot/2]
    cos, sin = ang.cos() * scaling, ang.sin() * scaling
    xr, xp = x[..., :rot], x[..., rot:]
    if interleaved:
        x0, x1 = xr[..., 0::2], xr[..., 1::2]
        r = torch.stack([x0 * cos - x1 * sin, x1 * cos + x0 * sin], -1).flatten(-2)
    else:
        x0, x1 = xr[..., : rot // 2], xr[..., rot // 2:]
        r = torch.cat([x0 * cos - x1 * sin, x1 * cos + x0 * sin], -1)
    return torch.cat([r, xp], -1)


_ACTS = {"silu": F.silu, "swish": F.silu, "gelu": F.gelu, "relu": F.relu,
         "gelu_pytorch_tanh": lambda x: F.gelu(x, approximate="tanh"), "gelu_tanh": lambda x: F.gelu(x, approximate="tanh"),
         "gelu_new": lambda x: F.gelu(x, approximate="tanh")}


@torch.no_grad()
def reference_forward(dc: DecoderConfig, sd: StateDict, ids: torch.Tensor, past=None, linear=None):
    """Logits [B, S, V] (fp32) and the updated KV list.  ``linear(name, x)`` may replace the dense products (the CPU
    low-bit path plugs its quantized matmul in here); ``past`` is a list of (k, v) per layer."""
    def lin(name, x, rows=None):
        """``rows`` = (first, count): the product with that row block of a stacked (per-expert) weight"""
        if linear is not None:
            return linear(name, x) if rows is None else linear(name, x, rows)
        w = sd[name] if rows is None else sd[name][rows[0]:rows[0] + rows[1]]
        y = x @ w.float().t()
        b = sd.get(name + ".bias")
        return y if b is None else y + b.float()

    def norm(x, name):
        w, nb = sd[name], sd.get(name + ".bias")
        if dc.norm == "layer":
            return F.layer_norm(x, (x.shape[-1],), w.float(), None if nb is None else nb.float(), dc.rms_eps)
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + dc.rms_eps) * (w.float() + dc.norm_offset)

    def mlp(i, x):
        if f"L{i}.router" in sd:
            t = x.reshape(-1, x.shape[-1])
            logits = t @ sd[f"L{i}.router"].float().t()
            p = torch.sigmoid(logits) if dc.moe_score == "sigmoid" else torch.softmax(logits, -1)
            choice = p + sd[f"L{i}.router_bias"].float() if f"L{i}.router_bias" in sd else p
            if dc.moe_n_group > 1:                  # only the experts of the best groups compete
                grouped = choice.view(choice.shape[0], dc.moe_n_group, -1)
                gs = grouped.topk(2, dim=-1).values.sum(-1) if dc.moe_group_score == "top2sum" else grouped.amax(-1)
                keep = torch.zeros_like(gs).scatter_(1, gs.topk(dc.moe_topk_group, dim=-1).indices, 1.0)
                choice = choice.masked_fill(keep.repeat_interleave(dc.num_experts // dc.moe_n_group, 1) == 0, 0.0)
            idx = torch.topk(choice, dc.top_k, dim=-1).indices
            w = p.gather(1, idx)
            if dc.norm_topk_prob:
                w = w / (w.sum(-1, keepdim=True) + 1e-20)
            e, inter, hd = dc.num_experts, dc.moe_intermediate_size, t.shape[-1]
            y = torch.zeros_like(t)
            for j in range(e):
                tok, slot = torch.where(idx == j)
                if tok.numel() == 0:
                    continue
                xe = t[tok]
                a = act(lin(f"L{i}.experts.gate", xe, rows=(j * inter, inter))) * lin(f"L{i}.experts.up", xe, rows=(j * inter, inter))
                y.index_add_(0, tok, lin(f"L{i}.experts.down", a, rows=(j * hd, hd)) * w[tok, slot, None])
            if f"L{i}.shared.gate" in sd:
                y = y + lin(f"L{i}.shared.down", act(lin(f"L{i}.shared.gate", t)) * lin(f"L{i}.shared.up", t))
            return y.reshape(x.shape)
        if dc.gated_mlp:
            return lin(f"L{i}.down", act(lin(f"L{i}.gate", x)) * lin(f"L{i}.up", x))
        return lin(f"L{i}.down", act(lin(f"L{i}.up", x)))

    b, s = ids.shape
    hq, hkv, d = dc.num_heads, dc.num_kv_heads, dc.head_dim
    rot = dc.rot_dim or d
    inv, scaling = compute_inv_freq(dc)
    p0 = 0 if not past else past[0][0].shape[2]
    pos = torch.arange(p0, p0 + s)
    h = sd["embed"][ids].float() * dc.embed_scale
    if "pos_embed" in sd:
        h = h + sd["pos_embed"][pos + dc.pos_offset].float()
    if "embed_norm" in sd:
        h = norm(h, "embed_norm")
    slopes = None
    if dc.alibi:
        slopes = torch.tensor(dc.alibi_slopes, dtype=torch.float32) if dc.alibi_slopes else alibi_slopes(hq)
    act = _ACTS[dc.act]
    scale = dc.attn_scale if dc.attn_scale is not None else 1.0 / math.sqrt(d)
    new_past = []
    for i in range(dc.num_layers):
        x = norm(h, f"L{i}.input_norm")
        if dc.mla_kv_lora:
            def rms(t, name):
                return t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + dc.rms_eps) * sd[name].float()
            ckv = lin(f"L{i}.kv_a", x)                 # [q latent (if any) | kv latent | rotary key]
            if dc.mla_q_lora:
                qh = lin(f"L{i}.q_b", rms(ckv[..., : dc.mla_q_lora], f"L{i}.q_a_norm"))
                ckv = ckv[..., dc.mla_q_lora:]
            else:
                qh = lin(f"L{i}.q", x)
            kvh = lin(f"L{i}.kv_b", rms(ckv[..., : dc.mla_kv_lora], f"L{i}.kv_a_norm"))
            kh = kvh[..., : hq * d].reshape(b, s, hq, d).clone()
            kh[..., :rot] = ckv[..., None, dc.mla_kv_lora:]          # the rotary key is shared by all heads
            qkv = torch.cat([qh, kh.reshape(b, s, hq * d), kvh[..., hq * d:]], -1)
        else:
            qkv = lin(f"L{i}.qkv", x)
        q = qkv[..., : hq * d].view(b, s, hq, d).transpose(1, 2)
        k = qkv[..., hq * d:(hq + hkv) * d].view(b, s, hkv, d).transpose(1, 2)
        v = qkv[..., (hq + hkv) * d:].view(b, s, hkv, d).transpose(1, 2)
        if not (dc.alibi or dc.no_rope):
            q, k = _rope(q, pos, inv, rot, dc.rope_interleaved, scaling), _rope(k, pos, inv, rot, dc.rope_interleaved, scaling)
        if past:
            k, v = torch.cat([past[i][0], k], 2), torch.cat([past[i][1], v], 2)
        new_past.append((k, v))
        kk, vv = k.repeat_interleave(hq // hkv, 1), v.repeat_interleave(hq // hkv, 1)
        sc = q @ kk.transpose(-1, -2) * scale
        L = k.shape[2]
        rel = torch.arange(L)[None] - (p0 + torch.arange(s))[:, None]
        if dc.alibi:
            sc = sc + slopes[None, :, None, None] * rel.float()
        sc = sc.masked_fill(rel > 0, float("-inf"))
        if dc.sliding_window:
            sc = sc.masked_fill(rel <= -dc.sliding_window, float("-inf"))
        a = (torch.softmax(sc, -1) @ vv).transpose(1, 2).reshape(b, s, hq * d)
        if dc.parallel_residual:
            x2 = norm(h, f"L{i}.post_norm") if f"L{i}.post_norm" in sd else x
            h = h + lin(f"L{i}.o", a) + mlp(i, x2)
            continue
        h = h + lin(f"L{i}.o", a)
        h = h + mlp(i, norm(h, f"L{i}.post_norm"))
    logits = lin("lm_head", norm(h, "final_norm")) * dc.logit_scale
    return logits, new_past


# ------------------------------------------------------------------------------------------------
# canonical form -> engine / CPU model
# ------------------------------------------------------------------------------------------------
LINEAR_SUFFIXES = (".qkv", ".o", ".gate", ".up", ".down", ".q", ".q_b", ".kv_a", ".kv_b")


def quantize_canonical(sd: StateDict, qtype: int, lm_head_qtype: Optional[int] = None):
    """Quantize the linears of a canonical state dict into ``FP4Params`` (host layout); norms / embedding / biases
    stay dense.  Rows whose in_features do not fit the block size stay dense fp16."""
    from ..transformers.low_bit_linear import FP4Params
    out = {}
    for name, t in sd.items():
        is_lin = name.endswith(LINEAR_SUFFIXES) or name == "lm_head"
        qt = (lm_head_qtype or qtype) if name == "lm_head" else qtype
        if is_lin and qt not in (Q.FP16, Q.BF16) and t.shape[1] % 64 == 0:
            out[name] = FP4Params(t.detach().float().contiguous(), requires_grad=False, quantized=False, _shape=None,
                                  qtype=qt, in_features=t.shape[1]).to("cpu")
        else:
            out[name] = t.detach().to(torch.float16)
    return out


def build_engine(dc: DecoderConfig, qsd: dict, device="cuda"):
    """Canonical (quantized) tensors -> ``NativeDecoder`` on ``device``."""
    from ..transformers.low_bit_linear import FP4Params
    from .decoder import LayerWeights, Lin, NativeDecoder
    dev = torch.device(device)

    def lin(name):
        w = qsd[name]
        w = w.to(dev) if isinstance(w, FP4Params) else w.to(dev, torch.float16)
        b = qsd.get(name + ".bias")
        return Lin(w, None if b is None else b.to(dev))

    def vec(name):
        return qsd[name].to(dev, torch.float16).contiguous() if name in qsd else None

    def mla(i):
        if not dc.mla_kv_lora:
            return None
        from .decoder import MLAWeights
        return MLAWeights(a=lin(f"L{i}.kv_a"), kv_a_norm=vec(f"L{i}.kv_a_norm"), kv_b=lin(f"L{i}.kv_b"),
                          kv_lora=dc.mla_kv_lora, q_lora=dc.mla_q_lora, q=None if dc.mla_q_lora else lin(f"L{i}.q"),
                          q_a_norm=vec(f"L{i}.q_a_norm"), q_b=lin(f"L{i}.q_b") if dc.mla_q_lora else None)

    def moe(i):
        if f"L{i}.router" not in qsd:
            return None
        from .moe import MoEWeights
        e, inter, hid = dc.num_experts, dc.moe_intermediate_size, dc.hidden_size
        ws, qt = [], None
        for n, shape in (("gate", (e, inter, hid)), ("up", (e, inter, hid)), ("down", (e, hid, inter))):
            w = qsd[f"L{i}.experts.{n}"]
            if isinstance(w, FP4Params):
                w = w.to(dev)
                assert w.qlayout == "t16", "expert weights reach the device in the tiled layout"
                ws.append(w.data)
                qt = w.qtype
            else:
                ws.append(w.to(dev, torch.float16).view(shape).contiguous())
                qt = Q.FP16
        shared = tuple(lin(f"L{i}.shared.{n}") for n in ("gate", "up", "down")) if f"L{i}.shared.gate" in qsd else None
        mw = MoEWeights(qsd[f"L{i}.router"].to(dev), ws[0], ws[1], ws[2], qt, e, dc.top_k, hid, inter, act=dc.act,
                        normalize=dc.norm_topk_prob, shared=shared)
        mw.n_group, mw.topk_group = dc.moe_n_group, dc.moe_topk_group
        mw.score, mw.group_score = dc.moe_score, dc.moe_group_score
        if f"L{i}.router_bias" in qsd:
            mw.route_bias = qsd[f"L{i}.router_bias"].to(dev, torch.float32)
        return mw

    def opt_lin(name):
        return lin(name) if name in qsd else None

    layers = [LayerWeights(input_norm=vec(f"L{i}.input_norm"), qkv=None if dc.mla_kv_lora else lin(f"L{i}.qkv"),
                           mla=mla(i), moe=moe(i), o=lin(f"L{i}.o"),
                           post_norm=vec(f"L{i}.post_norm"), gate=opt_lin(f"L{i}.gate") if dc.gated_mlp else None,
                           up=opt_lin(f"L{i}.up"), down=opt_lin(f"L{i}.down"), input_norm_b=vec(f"L{i}.input_norm.bias"),
                           post_norm_b=vec(f"L{i}.post_norm.bias"), window=dc.sliding_window)
              for i in range(dc.num_layers)]
    embed_norm = (vec("embed_norm"), vec("embed_norm.bias")) if "embed_norm" in qsd else None
    return NativeDecoder(dc, vec("embed"), layers, vec("final_norm"), lin("lm_head"), device=dev,
                         final_norm_b=vec("final_norm.bias"), pos_embed=vec("pos_embed"), embed_norm=embed_norm)


class NativeCausalLM:
    """A remote-code family loaded without its remote code: canonical low-bit tensors + ``DecoderConfig``.
    ``.to("cuda")`` builds the hipGraph engine; on the CPU ``generate`` runs the plain forward over the quantized
    matmul.  ``save_low_bit`` / ``load_low_bit`` keep the canonical tensors (host block layout) and the config."""

    def __init__(self, dc: DecoderConfig, qsd: dict, low_bit: str, source_config: Optional[dict] = None):
        self.dc, self.qsd, self.low_bit = dc, qsd, low_bit
        self.source_config = source_config or {}
        self.engine = None
        self.device = torch.device("cpu")
        self.first_cost = self.rest_cost_mean = self.peak_memory = 0.0

    # -- construction
    @classmethod
    def from_state_dict(cls, cfg: dict, sd: StateDict, low_bit="sym_int4"):
        mt = cfg.get("model_type")
        if mt not in ARCH_CONVERTERS:
            raise ValueError(f"no native weight map for model_type {mt!r}")
        dc, canon = pad_head_dim(*ARCH_CONVERTERS[mt](cfg, sd))
        return cls(dc, quantize_canonical(canon, Q.ggml_tensor_qtype[low_bit]), low_bit, cfg)

    @classmethod
    def from_pretrained(cls, path: str, low_bit="sym_int4"):
        cfg = json.load(open(os.path.join(path, "config.json")))
        return cls.from_state_dict(cfg, load_checkpoint_tensors(path), low_bit)

    # -- device
    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda":
            self.engine = build_engine(self.dc, self.qsd, device)
        else:
            self.engine = None
        self.device = device
        return self

    def half(self):
        return self

    def eval(self):
        return self

    # -- inference
    def _cpu_linear(self, name, x, rows=None):
        from ..transformers.low_bit_linear import FP4Params, lowbit_matmul
        w = self.qsd[name]
        if rows is not None:                               # one expert's block of a stacked weight
            if isinstance(w, FP4Params):
                from ..parallel.tensor_parallel import slice_param
                w = slice_param(w, rows=(rows[0], rows[0] + rows[1]))
            else:
                w = w[rows[0]:rows[0] + rows[1]]
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        y = lowbit_matmul(x2, w, w._shape[0]) if isinstance(w, FP4Params) else x2 @ w.float().t()
        b = self.qsd.get(name + ".bias")
        if b is not None:
            y = y + b.float()
        return y.view(*shp[:-1], -1)

    @torch.no_grad()
    def __call__(self, input_ids, past_key_values=None):
        if self.engine is not None:
            eng = self.engine
            eng.ensure_cache(input_ids.shape[0], input_ids.shape[1] + 1)
            eng.cache.reset()
            return eng.prefill(input_ids.to(self.device), last_only=False), None
        return reference_forward(self.dc, self.qsd, input_ids, past_key_values, linear=self._cpu_linear)

    @torch.no_grad()
    def generate(self, input_ids, max_new_tokens=32, do_sample=False, eos_token_id=None, **kw):
        if self.engine is not None:
            from ..transformers.utils import engine_kv_mode
            out = self.engine.generate(input_ids, max_new_tokens=max_new_tokens, do_sample=do_sample,
                                       eos_token_id=eos_token_id, kv_fp8=engine_kv_mode(), **kw)
            self.first_cost, self.rest_cost_mean = self.engine.first_cost, self.engine.rest_cost_mean
            self.peak_memory = self.engine.peak_memory
            return out
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        ids, past, cur = input_ids, None, input_ids
        for _ in range(max_new_tokens):
            logits, past = self(cur, past)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
            ids = torch.cat([ids, nxt], 1)
            cur = nxt
            if eos and all(int(t) in eos for t in nxt.view(-1)):
                break
        return ids

    # -- checkpoint
    def save_low_bit(self, path: str):
        from ..transformers.low_bit_linear import FP4Params
        os.makedirs(path, exist_ok=True)
        tensors, meta = {}, {}
        for name, t in self.qsd.items():
            if isinstance(t, FP4Params):
                data = t.data if t.data.device.type == "cpu" else t.to("cpu").data
                tensors[name] = data.view(torch.uint8) if data.dtype != torch.uint8 else data
                meta[name] = {"qtype": Q.qtype_name(t.qtype), "shape": list(t._shape)}
            else:
                tensors[name] = t.cpu()
        torch.save(tensors, os.path.join(path, "pytorch_model.bin"))
        cfg = dict(self.source_config)
        cfg["bigdl_transformers_low_bit"] = self.low_bit
        cfg["bigdl_native_arch"] = {"quantized": meta}
        json.dump(cfg, open(os.path.join(path, "config.json"), "w"), indent=1)
        json.dump({"all_checkpoint_keys": sorted(tensors)}, open(os.path.join(path, "load_keys.json"), "w"))

    @classmethod
    def load_low_bit(cls, path: str):
        from ..transformers.low_bit_linear import FP4Params
        cfg = json.load(open(os.path.join(path, "config.json")))
        if "bigdl_native_arch" not in cfg:
            raise ValueError(f"{path} is not a native-architecture low-bit checkpoint")
        meta = cfg.pop("bigdl_native_arch")["quantized"]
        low_bit = cfg.pop("bigdl_transformers_low_bit")
        tensors = torch.load(os.path.join(path, "pytorch_model.bin"), weights_only=True)
        dc = arch_config_only(cfg, padded=True)
        qsd = {}
        for name, t in tensors.items():
            if name in meta:
                qsd[name] = FP4Params(t, requires_grad=False, quantized=True, _shape=tuple(meta[name]["shape"]),
                                      qtype=Q.ggml_tensor_qtype[meta[name]["qtype"]])
            else:
                qsd[name] = t
        return cls(dc, qsd, low_bit, cfg)


def arch_config_only(cfg: dict, padded: bool = False) -> DecoderConfig:
    """DecoderConfig of a family from its config dict alone (the converters only read shapes from the tensors);
    ``padded``: as ``NativeCausalLM`` holds it, with an odd head_dim brought to a kernel size."""
    if padded:
        return pad_head_dim(arch_config_only(cfg), {})[0] if _needs_pad(arch_config_only(cfg)) else arch_config_only(cfg)
    mt = cfg.get("model_type")
    probe = {"chatglm": _probe_chatglm, "minicpm": _probe_minicpm, "minicpm3": _probe_minicpm3, "deepseek_v2": _probe_deepseek_v2, "deepseek_v3": _probe_deepseek_v3}.get(mt)
    if probe is not None:
        return probe(cfg)
    if mt in LN_CONFIGS:
        return LN_CONFIGS[mt](cfg)
    if mt == "baichuan":
        return _base_config(cfg, "baichuan", alibi="max_position_embeddings" not in cfg)
    if mt == "deci":
        return _cfg_decilm(cfg)
    if mt == "qwen":
        return _base_config(cfg, "qwen", intermediate_size=cfg["intermediate_size"] // 2,
                            head_dim=_get(cfg, "kv_channels", default=cfg["hidden_size"] // cfg["num_attention_heads"]))
    return _base_config(cfg, mt)


def _needs_pad(dc):
    return dc.head_dim not in ENGINE_HEAD_DIMS and dc.head_dim < max(ENGINE_HEAD_DIMS)


def _probe_chatglm(cfg):
    heads = cfg["num_attention_heads"]
    d = _get(cfg, "kv_channels", default=cfg["hidden_size"] // heads)
    mqa = bool(_get(cfg, "multi_query_attention", default=False))
    return _base_config(cfg, "chatglm", num_kv_heads=_get(cfg, "multi_query_group_num", default=heads) if mqa else heads,
                        head_dim=d, rot_dim=d // 2, rope_interleaved=True,
                        rope_theta=10000.0 * float(_get(cfg, "rope_ratio", default=1.0)),
                        intermediate_size=cfg["ffn_hidden_size"], vocab_size=_get(cfg, "padded_vocab_size", "vocab_size"))


def _probe_minicpm3(cfg):
    n, r, vd, d = _mla_layout(cfg)
    heads = cfg["num_attention_heads"]
    dc = _base_config(cfg, "minicpm3", tie_word_embeddings=True, num_kv_heads=heads, head_dim=d, rot_dim=r,
                      rope_interleaved=True,      # the weight map puts the rotate-half pairs side by side
                      attn_scale=1.0 / math.sqrt(n + r), mla_q_lora=int(cfg.get("q_lora_rank") or 0),
                      mla_kv_lora=int(cfg["kv_lora_rank"]))
    dc.embed_scale = float(_get(cfg, "scale_emb", default=1.0))
    dc.logit_scale = float(_get(cfg, "dim_model_base", default=dc.hidden_size)) / dc.hidden_size
    return dc


def _probe_minicpm(cfg):
    dc = _base_config(cfg, "minicpm", tie_word_embeddings=True)
    dc.embed_scale = float(_get(cfg, "scale_emb", default=1.0))
    dc.logit_scale = float(_get(cfg, "dim_model_base", default=dc.hidden_size)) / dc.hidden_size
    return dc


def load_checkpoint_tensors(path: str) -> StateDict:
    """All tensors of a HF checkpoint directory (safetensors preferred; ``.bin`` shards through the weights-only
    unpickler)."""
    import glob
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    sd: StateDict = {}
    if files:
        from safetensors.torch import load_file
        for f in files:
            sd.update(load_file(f))
        return sd
    files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not files:
        raise FileNotFoundError(f"no weight files under {path}")
    for f in files:
        sd.update(torch.load(f, map_location="cpu", weights_only=True))
    return sd


def is_native_arch(config_dict: dict, opt_in: Optional[bool] = None) -> bool:
    """True when the checkpoint's model_type has a weight map here and either HuggingFace has no built-in class for it
    or the caller opted into the native engine (``native_engine=True`` / BIGDL_NATIVE_ENGINE=1) for a family that also
    has a module path."""
    mt = config_dict.get("model_type")
    if mt not in ARCH_CONVERTERS:
        return False
    if opt_in is None:
        opt_in = os.environ.get("BIGDL_NATIVE_ENGINE") == "1"
    if opt_in:
        return True
    from transformers.models.auto.configuration_auto import CONFIG_MAPPING_NAMES
    return mt not in CONFIG_MAPPING_NAMES
