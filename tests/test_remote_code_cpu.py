"""Remote-code families on the module path (bigdl_amd/models/remote_code.py).  The families' own modelling files are not
available offline, so every test builds a stand-in module with the family's attribute protocol (class name, sub-module
and buffer names) next to a plain PyTorch statement of what that module computes, converts the stand-in with
``optimize_model`` and compares.  Parity with the real remote code is therefore unpinned beyond these protocols."""
import math

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from bigdl_amd.models import remote_code as rc
from bigdl_amd.optimize import optimize_model

H, NH, D = 64, 4, 16


def _causal(q, k, v):
    """q [B,Hq,S,D], k/v [B,Hkv,L,D]; the last query sees every key"""
    g = q.shape[1] // k.shape[1]
    k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
    s, L = q.shape[2], k.shape[2]
    sc = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    sc = sc.masked_fill(~torch.ones(s, L, dtype=torch.bool).tril(L - s), float("-inf"))
    return (sc.softmax(-1) @ v).transpose(1, 2).reshape(q.shape[0], s, -1)


class _Holder(nn.Module):
    """What ``optimize_model`` sees: a model with a config and the family module inside."""

    def __init__(self, inner, model_type, **cfg):
        super().__init__()
        self.inner = inner
        self.config = type("Cfg", (), dict(model_type=model_type, **cfg))()

    def forward(self, *a, **k):
        return self.inner(*a, **k)


# ---------------------------------------------------------------- phixtral
class _PhiMLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1, self.fc2, self.act = nn.Linear(H, 4 * H), nn.Linear(4 * H, H), nn.GELU(approximate="tanh")

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class MoE(nn.Module):
    def __init__(self, n=4, k=2):
        super().__init__()
        self.mlp = nn.ModuleList([_PhiMLP() for _ in range(n)])
        self.gate = nn.Linear(H, n, bias=False)
        self.num_experts_per_tok = k

    def forward(self, x):       # the family's own (dense) statement: every expert on every token, masked by the routing
        p = F.softmax(self.gate(x), -1, dtype=torch.float)
        w, idx = p.topk(self.num_experts_per_tok, -1)
        w = w / w.sum(-1, keepdim=True)
        out = torch.zeros_like(x)
        for e, m in enumerate(self.mlp):
            out = out + m(x) * (w * (idx == e)).sum(-1, keepdim=True).to(x.dtype)
        return out


@pytest.mark.parametrize("rows", [1, 7])
def test_phixtral_moe_block(rows):
    torch.manual_seed(0)
    moe = MoE().eval()
    x = torch.randn(1, rows, H)
    want = moe(x)
    assert rc.bind_remote_code_forwards(moe) == 1
    torch.testing.assert_close(moe(x), want, rtol=1e-5, atol=1e-5)
    # converted: experts are low-bit linears, the rebound forward runs only the chosen ones
    q = optimize_model(_Holder(MoE().eval(), "phi-msft"), low_bit="sym_int8")
    q.inner.load_state_dict({k: v for k, v in q.inner.state_dict().items()})
    assert type(q.inner.mlp[0].fc1).__name__ == "LowBitLinear" and q.inner.forward.__func__ is rc.phixtral_moeblock_forward
    calls = []
    for e, m in enumerate(q.inner.mlp):
        m.register_forward_hook(lambda *_a, e=e: calls.append(e))
    q.inner(x[:, :1])
    assert len(calls) == 2


# ---------------------------------------------------------------- Baichuan-M1
class BaichuanAttention(nn.Module):
    def __init__(self, hkv=2):
        super().__init__()
        self.num_heads, self.num_key_value_heads, self.head_dim, self.layer_idx, self.hidden_size = NH, hkv, D, 0, H
        self.W_pack = nn.Linear(H, (NH + 2 * hkv) * D, bias=False)
        self.o_proj = nn.Linear(NH * D, H, bias=False)
        self.conv_k = nn.Parameter(torch.randn(1, 1, hkv, 1, 2) * 0.5)
        self.conv_v = nn.Parameter(torch.randn(1, 1, hkv, 1, 2) * 0.5)
        self.rotary_emb = type("R", (), {})()
        self.rotary_emb.inv_freq = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))

    def reference(self, x):
        """whole sequence at once: unfold-style convolution over the sequence with a zero before the first token"""
        b, s, _ = x.shape
        hq, hkv = self.num_heads, self.num_key_value_heads
        q, k, v = self.W_pack(x).view(b, s, hq + 2 * hkv, D).split([hq, hkv, hkv], 2)

        def conv(u, taps):
            pad = F.pad(u, (0, 0, 0, 0, 1, 0))
            win = pad.unfold(1, 2, 1)                                  # [b, s, h, d, 2]
            return (win * taps).sum(-1)
        k, v = conv(k, self.conv_k), conv(v, self.conv_v)
        ang = torch.arange(s)[:, None].float() * self.rotary_emb.inv_freq[None]
        cos, sin = torch.cat([ang.cos()] * 2, -1)[None, :, None], torch.cat([ang.sin()] * 2, -1)[None, :, None]
        rot = lambda t: t * cos + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * sin      # noqa: E731
        return self.o_proj(_causal(rot(q).transpose(1, 2), rot(k).transpose(1, 2), v.transpose(1, 2)))


def test_baichuan_m1_attention_prefill_then_decode():
    from bigdl_amd.transformers.kv import DynamicNormalCache
    torch.manual_seed(1)
    att = BaichuanAttention().eval()
    x = torch.randn(2, 9, H)
    with torch.no_grad():
        want = att.reference(x)
        assert rc.bind_remote_code_forwards(att) == 1
        cache = DynamicNormalCache()
        got, _, _ = att(x[:, :6], past_key_value=cache, use_cache=True)
        torch.testing.assert_close(got, want[:, :6], rtol=1e-4, atol=1e-5)
        for t in range(6, 9):       # token by token: the filter needs the previous token's raw key / value
            step, _, _ = att(x[:, t:t + 1], past_key_value=cache, use_cache=True,
                             position_ids=torch.tensor([[t]]).expand(2, 1))
            torch.testing.assert_close(step[:, 0], want[:, t], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- ChatGLM-6B (v1)
class SelfAttention(nn.Module):
    def __init__(self, two_d=True):
        super().__init__()
        self.num_attention_heads_per_partition, self.hidden_size_per_attention_head = NH, D
        self.position_encoding_2d = two_d
        self.query_key_value = nn.Linear(H, 3 * NH * D)
        self.dense = nn.Linear(NH * D, H)
        self.rotary_emb = type("R", (), {"base": 10000.0})()

    def reference(self, x, position_ids):
        """x [S, B, H]; the family's statement: per-head [q | k | v], rotary with positions on the first half of each head
        and block positions on the second half"""
        s, b, _ = x.shape
        q, k, v = self.query_key_value(x).view(s, b, NH, 3 * D).split(D, -1)

        def rot(t, pos):
            dim = t.shape[-1]
            inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2).float() / dim))
            ang = pos.transpose(0, 1)[..., None].float() * inv                  # [S, B, dim/2]
            cos, sin = torch.cat([ang.cos()] * 2, -1)[:, :, None], torch.cat([ang.sin()] * 2, -1)[:, :, None]
            return t * cos + torch.cat([-t[..., dim // 2:], t[..., :dim // 2]], -1) * sin
        if self.position_encoding_2d:
            p, bp = position_ids[:, 0], position_ids[:, 1]
            q = torch.cat([rot(q[..., :D // 2], p), rot(q[..., D // 2:], bp)], -1)
            k = torch.cat([rot(k[..., :D // 2], p), rot(k[..., D // 2:], bp)], -1)
        else:
            q, k = rot(q, position_ids), rot(k, position_ids)
        out = _causal(q.permute(1, 2, 0, 3), k.permute(1, 2, 0, 3), v.permute(1, 2, 0, 3))
        return self.dense(out.transpose(0, 1))


@pytest.mark.parametrize("two_d", [True, False])
def test_chatglm_v1_attention(two_d):
    torch.manual_seed(2)
    att = SelfAttention(two_d).eval()
    s, b = 8, 2
    x = torch.randn(s, b, H)
    if two_d:       # a 5-token context then generation: positions stop at the mask slot, block positions count on
        pos = torch.tensor([0, 1, 2, 3, 4, 4, 4, 4])
        blk = torch.tensor([0, 0, 0, 0, 0, 1, 2, 3])
        position_ids = torch.stack([pos, blk])[None].expand(b, 2, s)
    else:
        position_ids = torch.arange(s)[None].expand(b, s)
    with torch.no_grad():
        want = att.reference(x, position_ids)
        assert rc.bind_remote_code_forwards(att) == 1
        out, present = att(x[:5], position_ids[..., :5], torch.zeros(1, 1, 1, 1, dtype=torch.bool), 0, use_cache=True)
        torch.testing.assert_close(out, want[:5], rtol=1e-4, atol=1e-5)
        for t in range(5, s):
            out, present = att(x[t:t + 1], position_ids[..., t:t + 1], torch.zeros(1, 1, 1, 1, dtype=torch.bool), 0,
                               layer_past=present, use_cache=True)
            torch.testing.assert_close(out[0], want[t], rtol=1e-4, atol=1e-5)
        assert present[0].shape == (s, b, NH, D)


# ---------------------------------------------------------------- Qwen-VL (v1)
class QWenAttention(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_heads, self.head_dim, self.split_size = NH, D, NH * D
        self.c_attn, self.c_proj = nn.Linear(H, 3 * NH * D), nn.Linear(NH * D, H, bias=False)
        self.use_logn_attn = True
        self.logn_tensor = torch.tensor([math.log(i, 6) if i > 6 else 1.0 for i in range(1, 65)])[None, :, None, None]

    def reference(self, x, cos, sin):
        b, s, _ = x.shape
        q, k, v = (t.view(b, s, NH, D) for t in self.c_attn(x).split(self.split_size, 2))
        rot = lambda t: t * cos + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * sin      # noqa: E731
        q, k = rot(q) * self.logn_tensor[:, :s], rot(k)
        return self.c_proj(_causal(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)))


def test_qwen_vl_text_attention_with_cache_and_logn():
    torch.manual_seed(3)
    att = QWenAttention().eval()
    model = _Holder(att, "qwen", visual={"image_size": 448})
    s = 10
    x = torch.randn(1, s, H)
    ang = torch.arange(s)[:, None].float() * (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D)))[None]
    cos, sin = torch.cat([ang.cos()] * 2, -1)[None, :, None], torch.cat([ang.sin()] * 2, -1)[None, :, None]
    with torch.no_grad():
        want = att.reference(x, cos, sin)
        assert rc.bind_remote_code_forwards(model) == 1
        out, present = att(x[:, :7], rotary_pos_emb=[cos[:, :7], sin[:, :7]], use_cache=True)
        torch.testing.assert_close(out, want[:, :7], rtol=1e-4, atol=1e-5)
        for t in range(7, s):
            out, present = att(x[:, t:t + 1], rotary_pos_emb=[cos[:, :t + 1], sin[:, :t + 1]], layer_past=present, use_cache=True)
            torch.testing.assert_close(out[:, 0], want[:, t], rtol=1e-4, atol=1e-5)


class Resampler(nn.Module):
    def __init__(self, grid=2, src=4):
        super().__init__()
        self.query = nn.Parameter(torch.randn(grid * grid, H) * 0.1)
        self.pos_embed = nn.Parameter(torch.randn(grid * grid, H) * 0.1)
        self.kv_proj, self.ln_q, self.ln_kv = nn.Linear(32, H, bias=False), nn.LayerNorm(H), nn.LayerNorm(H)
        self.attn = nn.MultiheadAttention(H, NH)


def test_qwen_vl_resampler_pools_image_tokens():
    torch.manual_seed(4)
    r = Resampler().eval()
    x = torch.randn(3, 4, 32)                     # 3 images, 2 x 2 patches: the position table fits without resizing
    with torch.no_grad():
        kv = r.ln_kv(r.kv_proj(x)).permute(1, 0, 2)
        q = r.ln_q(r.query)[:, None].repeat(1, 3, 1)
        want = r.attn(q + r.pos_embed[:, None], kv + r.pos_embed[:, None], kv)[0].permute(1, 0, 2)
        assert rc.bind_remote_code_forwards(r) == 1
        torch.testing.assert_close(r(x), want, rtol=1e-5, atol=1e-6)
        assert r(torch.randn(1, 16, 32)).shape == (1, 4, H)      # 4 x 4 patches: the 2 x 2 table is resized (bicubic)


# ---------------------------------------------------------------- InternLM-XComposer2
class PLoRA(nn.Linear):
    def __init__(self, i, o, r=4):
        super().__init__(i, o, bias=False)
        self.Plora_A, self.Plora_B = nn.Linear(i, r, bias=False), nn.Linear(r, o, bias=False)
        self.lora_dropout, self.lora_scaling = nn.Dropout(0.0), 0.5

    def forward(self, x, im_mask=None):
        y = super().forward(x)
        if im_mask is not None and im_mask.any():
            y = y.clone()
            y[im_mask] += self.Plora_B(self.Plora_A(x[im_mask])) * self.lora_scaling
        return y


class InternLM2MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.w1, self.w3, self.w2 = PLoRA(H, 96), PLoRA(H, 96), PLoRA(96, H)

    def forward(self, x, im_mask):
        return self.w2(F.silu(self.w1(x, im_mask)) * self.w3(x, im_mask), im_mask)


class InternLM2Attention(nn.Module):
    def __init__(self, hkv=2):
        super().__init__()
        self.num_heads, self.num_key_value_heads, self.head_dim = NH, hkv, D
        self.wqkv, self.wo = PLoRA(H, (NH + 2 * hkv) * D), PLoRA(NH * D, H)
        self.rotary_emb = type("R", (), {"base": 10000.0})()

    def forward(self, x, im_mask):
        b, s, _ = x.shape
        g = NH // self.num_key_value_heads
        qkv = self.wqkv(x, im_mask).view(b, s, self.num_key_value_heads, g + 2, D)
        q = qkv[..., :g, :].reshape(b, s, NH, D)
        k, v = qkv[..., g, :], qkv[..., g + 1, :]
        ang = torch.arange(s)[:, None].float() * (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D)))[None]
        cos, sin = torch.cat([ang.cos()] * 2, -1)[None, :, None], torch.cat([ang.sin()] * 2, -1)[None, :, None]
        rot = lambda t: t * cos + torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1) * sin      # noqa: E731
        return self.wo(_causal(rot(q).transpose(1, 2), rot(k).transpose(1, 2), v.transpose(1, 2)), im_mask)


def test_xcomposer2_partial_lora_survives_low_bit_conversion():
    torch.manual_seed(5)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.attention, self.feed_forward = InternLM2Attention(), InternLM2MLP()
    blk = Block().eval()
    x = torch.randn(1, 9, H)
    im = torch.zeros(1, 9, dtype=torch.bool)
    im[0, 2:6] = True
    with torch.no_grad():
        want_a, want_m = blk.attention(x, im), blk.feed_forward(x, im)
        want_plain = blk.feed_forward(x, None)
    q = optimize_model(_Holder(blk, "internlmxcomposer2"), low_bit="sym_int8").inner
    assert type(q.feed_forward.w1).__name__ == "LowBitLinear" and type(q.feed_forward.w1_Plora_A) is nn.Linear
    with torch.no_grad():
        got_a, _, _ = q.attention(x, im_mask=im)
        got_m = q.feed_forward(x, im_mask=im)
        got_ranges = q.feed_forward(x, im_mask=[(2, 6)])
        got_plain = q.feed_forward(x, im_mask=[])
    rel = lambda a, b: ((a - b).abs().mean() / b.abs().mean()).item()      # noqa: E731
    assert rel(got_a, want_a) < 0.03 and rel(got_m, want_m) < 0.03 and rel(got_plain, want_plain) < 0.03
    torch.testing.assert_close(got_ranges, got_m)
    assert rel(got_m, want_plain) > 2 * rel(got_m, want_m)          # the adapter really contributes on the image rows


# ---------------------------------------------------------------- Janus vision attention, MiniCPM-V wrapper
class Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_heads, self.head_dim = NH, D
        self.qkv, self.proj, self.proj_drop = nn.Linear(H, 3 * H), nn.Linear(H, H), nn.Identity()
        self.q_norm, self.k_norm = nn.LayerNorm(D), nn.LayerNorm(D)


def test_janus_vision_attention():
    torch.manual_seed(6)
    att = Attention().eval()
    model = _Holder(att, "multi_modality")
    type(model).__name__ = "JanusHolder"
    x = torch.randn(2, 10, H)
    with torch.no_grad():
        q, k, v = att.qkv(x).view(2, 10, 3, NH, D).permute(2, 0, 3, 1, 4).unbind(0)
        sc = (att.q_norm(q) @ att.k_norm(k).transpose(-1, -2) / math.sqrt(D)).softmax(-1)
        want = att.proj((sc @ v).transpose(1, 2).reshape(2, 10, H))
        assert rc.bind_remote_code_forwards(model) == 1
        torch.testing.assert_close(att(x), want, rtol=1e-4, atol=1e-5)


def test_minicpmv_generate_wrapper_and_penalty_processor():
    seen = {}

    class MiniCPMV(nn.Module):
        def __init__(self):
            super().__init__()
            self.llm = nn.Linear(2, 2)

        def generate(self, *a, **kw):
            seen.update(kw)
            return "ok"
    m = MiniCPMV()
    assert rc.bind_remote_code_forwards(m) == 1 and rc.bind_remote_code_forwards(m) == 1
    assert m.generate(repetition_penalty=1.3, max_new_tokens=4) == "ok"
    assert "repetition_penalty" not in seen and seen["do_sample"] is False
    proc = seen["logits_processor"][0]
    ids = torch.tensor([[1, 3, 3]])
    scores = torch.tensor([[0.5, 2.0, -1.0, -2.0, 4.0]])
    want = scores.clone()
    want[0, 1], want[0, 3] = 2.0 / 1.3, -2.0 * 1.3
    torch.testing.assert_close(proc(ids, scores.clone()), want)


def test_wrapper_with_llm_and_speech_decoder_gets_both_optimized():
    """MiniCPM-o / Megrez-o style wrappers keep their text decoder under ``.llm`` and (MiniCPM-o) a second llama-style
    decoder under ``.tts.model`` (reference convert.py:1030-1047): both get the merged qkv projection and low-bit
    linears, and the wrapper's logits stay those of the dense model."""
    import copy
    from transformers import LlamaConfig, LlamaForCausalLM, LlamaModel, PretrainedConfig
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=96, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=64)

    class Speech(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = LlamaModel(cfg)

    class Omni(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = PretrainedConfig()
            self.config.model_type = "minicpmo"
            self.llm = LlamaForCausalLM(cfg)
            self.tts = Speech()

        def forward(self, ids):
            return self.llm(ids).logits, self.tts.model(ids).last_hidden_state

    m = Omni().eval()
    ids = torch.randint(3, 96, (1, 7))
    with torch.no_grad():
        want = m(ids)
    q = optimize_model(copy.deepcopy(m), low_bit="sym_int8")
    for attn in (q.llm.model.layers[0].self_attn, q.tts.model.layers[1].self_attn):
        assert isinstance(attn.qkv_proj, LowBitLinear) and not hasattr(attn, "q_proj")
    with torch.no_grad():
        got = q(ids)
    for a, b in zip(got, want):
        assert ((a.float() - b).abs().mean() / b.abs().mean()).item() < 0.05
