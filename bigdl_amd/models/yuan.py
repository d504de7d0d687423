"""Yuan 2.0 ("localized filtering-based attention") as a self-contained HuggingFace-style model.

The architecture is not part of ``transformers``; upstream checkpoints bring it as remote code, which is not available
offline, so GGUF files of this family (and plain checkpoints whose ``config.model_type`` is ``yuan``) are served by this
implementation.  Capability map to the reference: the vendored model the GGUF loader instantiates
(python/llm/src/ipex_llm/transformers/gguf/models/yuan2.py:26-130 with model_implement/yuan2/yuan_hf_model.py) and the
optimized forwards of python/llm/src/ipex_llm/transformers/models/yuan.py:50-97 (MLP), :99-250 (attention with the
two-token filter memory).

What the model computes, per decoder layer:

  f   = RMSNorm( conv2(conv1(x)) + x )        two causal 2-tap convolutions over time (C -> C/2 -> C): f_t sees x_t, x_{t-1}, x_{t-2}
  q,k = rope( split(q_proj(f), k_proj(f)) )   q and k come from the FILTERED stream; the split pairs up q / k head blocks
  v   = v_proj(x)                             values come from the unfiltered stream
  x   = x + o_proj(softmax(q k^T / sqrt(d)) v);   x = x + down( gate(x) * silu(up(x)) )     (activation on the *up* branch)

State for incremental decoding: keys, values and the last two layer inputs (the filter's memory).  The two convolutions
are evaluated as four [C, C/2]-shaped matrix products on the (t-2, t-1, t) window instead of ``conv2d`` calls on a
permuted 4-D view, so a decode step is four small GEMVs per layer and the weights keep the checkpoint's
``conv{1,2}.weight`` [out, in, 2, 1] layout.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn
from transformers import PretrainedConfig, PreTrainedModel
from transformers.modeling_outputs import CausalLMOutputWithPast


class YuanConfig(PretrainedConfig):
    model_type = "yuan"

    def __init__(self, vocab_size=135040, hidden_size=2048, intermediate_size=8192, num_hidden_layers=24,
                 num_attention_heads=32, hidden_act="silu", max_position_embeddings=8192, rms_norm_eps=1e-6,
                 rope_theta=10000.0, use_cache=True, pad_token_id=None, bos_token_id=None, eos_token_id=None,
                 tie_word_embeddings=False, **kwargs):
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.hidden_act, self.max_position_embeddings, self.rms_norm_eps = hidden_act, max_position_embeddings, rms_norm_eps
        self.rope_theta, self.use_cache = rope_theta, use_cache
        super().__init__(pad_token_id=pad_token_id, bos_token_id=bos_token_id, eos_token_id=eos_token_id,
                         tie_word_embeddings=tie_word_embeddings, **kwargs)


class YuanRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        h = x.float()
        h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return (self.weight.float() * h).to(x.dtype)


class LocalizedFiltering(nn.Module):
    """conv1 / conv2 keep the checkpoint's parameter shapes ([out, in, 2, 1] + bias); tap 0 multiplies the previous
    time step, tap 1 the current one."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.conv1 = nn.Conv2d(hidden_size, hidden_size // 2, (2, 1), padding=(1, 0))
        self.conv2 = nn.Conv2d(hidden_size // 2, hidden_size, (2, 1), padding=(1, 0))
        self.output_layernorm = YuanRMSNorm(hidden_size, eps)

    @staticmethod
    def _taps(conv, seq):
        """seq [B, T + 1, Cin] (one step of history in front) -> [B, T, Cout]"""
        w = conv.weight
        return F.linear(seq[:, :-1], w[:, :, 0, 0]) + F.linear(seq[:, 1:], w[:, :, 1, 0], conv.bias)

    def forward(self, x, before: Optional[torch.Tensor] = None):
        """x [B, T, C]; ``before`` [B, 2, C]: the two inputs preceding x (None at the start of a sequence, where the
        convolutions see zero padding)."""
        b, t, c = x.shape
        hist = before if before is not None else x.new_zeros(b, 2, c)
        o1 = self._taps(self.conv1, torch.cat([hist, x], 1))             # positions -1 .. T-1
        if before is None:
            o1 = torch.cat([torch.zeros_like(o1[:, :1]), o1[:, 1:]], 1)  # conv2's own padding: a zero, not conv1(0) + b
        o2 = self._taps(self.conv2, o1)                                   # positions 0 .. T-1
        return self.output_layernorm(o2 + x)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), -1)


class YuanAttention(nn.Module):
    def __init__(self, config: YuanConfig, layer_idx: int):
        super().__init__()
        c = config.hidden_size
        self.layer_idx = layer_idx
        self.num_heads = config.num_attention_heads
        self.head_dim = c // self.num_heads
        self.lf_gate = LocalizedFiltering(c, config.rms_norm_eps)
        self.q_proj = nn.Linear(c, c, bias=False)
        self.k_proj = nn.Linear(c, c, bias=False)
        self.v_proj = nn.Linear(c, c, bias=False)
        self.o_proj = nn.Linear(c, c, bias=False)
        inv = 1.0 / (config.rope_theta ** (torch.arange(0, self.head_dim, 2).float() / self.head_dim))
        self.register_buffer("inv_freq", inv, persistent=False)

    def forward(self, x, position_ids, past=None, attention_mask=None):
        b, t, c = x.shape
        h, d = self.num_heads, self.head_dim
        before = past[2] if past is not None else None
        mem = torch.cat([before[:, -1:], x], 1)[:, -2:] if before is not None else \
            (x[:, -2:] if t >= 2 else torch.cat([x.new_zeros(b, 1, c), x], 1))
        v = self.v_proj(x).view(b, t, h, d).transpose(1, 2)
        f = self.lf_gate(x, before)
        qk = torch.cat([self.q_proj(f), self.k_proj(f)], -1).view(b, t, h, 2 * d)
        q, k = (s.transpose(1, 2) for s in torch.chunk(qk, 2, -1))
        ang = position_ids[:, :, None].float() * self.inv_freq[None, None, :].float()        # [B, T, d/2]
        cos, sin = (torch.cat([fn(ang), fn(ang)], -1)[:, None].to(q.dtype) for fn in (torch.cos, torch.sin))
        q, k = q * cos + _rotate_half(q) * sin, k * cos + _rotate_half(k) * sin
        if past is not None:
            k, v = torch.cat([past[0], k], 2), torch.cat([past[1], v], 2)
        if x.device.type == "cuda" and attention_mask is None and not torch.is_grad_enabled() and d in (64, 96, 128, 256) \
                and x.dtype == torch.float16:
            from ..ops import hip
            o = hip.attn_prefill(q, k, v, scale=1.0 / math.sqrt(d), causal=True)          # [B, T, C]
            return self.o_proj(o), (k, v, mem)
        else:
            s = k.shape[2]
            bias = None
            if t > 1 or attention_mask is not None:
                keep = torch.ones(t, s, dtype=torch.bool, device=x.device).tril(s - t)[None, None]
                if attention_mask is not None:
                    keep = keep & attention_mask[:, None, None, :s].bool()
                bias = torch.zeros(keep.shape, dtype=q.dtype, device=x.device).masked_fill(~keep, torch.finfo(q.dtype).min)
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, scale=1.0 / math.sqrt(d))
        o = o.transpose(1, 2).reshape(b, t, c)
        return self.o_proj(o), (k, v, mem)


class YuanMLP(nn.Module):
    def __init__(self, config: YuanConfig):
        super().__init__()
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x):
        return self.down_proj(self.gate_proj(x) * F.silu(self.up_proj(x)))


class YuanDecoderLayer(nn.Module):
    def __init__(self, config: YuanConfig, layer_idx: int):
        super().__init__()
        self.self_attn = YuanAttention(config, layer_idx)
        self.mlp = YuanMLP(config)
        self.input_layernorm = YuanRMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = YuanRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, x, position_ids, past=None, attention_mask=None):
        a, present = self.self_attn(self.input_layernorm(x), position_ids, past, attention_mask)
        x = x + a
        return x + self.mlp(self.post_attention_layernorm(x)), present


class YuanModel(nn.Module):
    def __init__(self, config: YuanConfig):
        super().__init__()
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, config.pad_token_id)
        self.layers = nn.ModuleList([YuanDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = YuanRMSNorm(config.hidden_size, config.rms_norm_eps)


class YuanForCausalLM(PreTrainedModel):
    config_class = YuanConfig
    base_model_prefix = "model"
    _no_split_modules = ["YuanDecoderLayer"]

    def __init__(self, config: YuanConfig):
        super().__init__(config)
        self.model = YuanModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def forward(self, input_ids=None, past_key_values=None, attention_mask=None, position_ids=None, inputs_embeds=None,
                use_cache=None, labels=None, **unused):
        x = inputs_embeds if inputs_embeds is not None else self.model.embed_tokens(input_ids)
        b, t, _ = x.shape
        past_len = past_key_values[0][0].shape[2] if past_key_values else 0
        if position_ids is None:
            position_ids = torch.arange(past_len, past_len + t, device=x.device)[None].expand(b, t)
        if attention_mask is not None and bool(attention_mask.all()):
            attention_mask = None
        presents = []
        for i, layer in enumerate(self.model.layers):
            x, present = layer(x, position_ids, past_key_values[i] if past_key_values else None, attention_mask)
            presents.append(present)
        logits = self.lm_head(self.model.norm(x))
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits[:, :-1].float().reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1),
                                   ignore_index=-100)
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=tuple(presents) if use_cache is not False else None)

    @torch.no_grad()
    def generate(self, input_ids=None, max_new_tokens=32, do_sample=False, temperature=1.0, top_k=0, top_p=1.0,
                 eos_token_id=None, pad_token_id=None, streamer=None, attention_mask=None, **unused):
        """Greedy / sampled decoding on the model's own (keys, values, filter memory) state."""
        from .decoder import sample_logits
        input_ids = unused.pop("inputs", input_ids)
        eos = eos_token_id if eos_token_id is not None else self.config.eos_token_id
        eos = set([eos] if isinstance(eos, int) else (eos or []))
        pad = pad_token_id if pad_token_id is not None else (self.config.pad_token_id if self.config.pad_token_id is not None
                                                             else (next(iter(eos)) if eos else 0))
        out = self(input_ids, use_cache=True)
        past, ids = out.past_key_values, input_ids
        alive = torch.ones(ids.shape[0], dtype=torch.bool, device=ids.device)
        if streamer is not None:
            streamer.put(input_ids.cpu())
        for _ in range(max_new_tokens):
            logits = out.logits[:, -1].float()
            nxt = sample_logits(logits, temperature, top_k, top_p) if do_sample else logits.argmax(-1)
            nxt = torch.where(alive, nxt, torch.full_like(nxt, pad))
            ids = torch.cat([ids, nxt[:, None]], 1)
            if streamer is not None:
                streamer.put(nxt.cpu())
            if eos:
                alive = alive & ~torch.isin(nxt, torch.tensor(sorted(eos), device=nxt.device))
                if not bool(alive.any()):
                    break
            out = self(nxt[:, None], past_key_values=past, use_cache=True)
            past = out.past_key_values
        if streamer is not None:
            streamer.end()
        return ids


def register():
    """Make ``AutoConfig`` / ``AutoModelForCausalLM`` resolve ``model_type == "yuan"`` to the classes above (checkpoints of
    this family otherwise need their remote code)."""
    from transformers import AutoConfig, AutoModelForCausalLM
    try:
        AutoConfig.register("yuan", YuanConfig)
        AutoModelForCausalLM.register(YuanConfig, YuanForCausalLM)
    except ValueError:
        pass            # already registered (second import path, or the remote code got there first)
