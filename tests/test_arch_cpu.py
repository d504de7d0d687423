"""Remote-code families through their weight-name maps (bigdl_amd/models/arch.py).

Each family gets a random checkpoint in its ORIGINAL tensor naming and a small forward written the way the family's
own modeling file computes it (packed projections split / chunked / rearranged as there).  The canonical conversion
must reproduce those logits exactly in fp32, and the quantized CPU model must stay within the reference's 5 % bound
(python/llm/test/inference/test_transformers_api.py:151-169)."""
import json
import math

import pytest
import torch
import torch.nn.functional as F

from bigdl_amd.models import arch

V, H, L, NH, D = 320, 128, 2, 4, 32


def _w(*shape, s=0.05):
    return torch.randn(*shape) * s


def _norm(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _attend(q, k, v, bias=None):
    """q [B,S,Hq,D], k/v [B,S,Hkv,D] -> [B,S,Hq*D], causal; bias: additive [Hq, 1, S]"""
    b, s, hq, d = q.shape
    g = hq // k.shape[2]
    q, k, v = q.transpose(1, 2), k.transpose(1, 2).repeat_interleave(g, 1), v.transpose(1, 2).repeat_interleave(g, 1)
    sc = q @ k.transpose(-1, -2) / math.sqrt(d)
    if bias is not None:
        sc = sc + bias
    sc = sc.masked_fill(torch.triu(torch.ones(s, s, dtype=torch.bool), 1), float("-inf"))
    return (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(b, s, hq * d)


def _rope_half(x, theta=10000.0):
    b, s, h, d = x.shape
    inv = 1.0 / (theta ** (torch.arange(0, d, 2).float() / d))
    ang = torch.arange(s)[:, None].float() * inv[None]
    cos, sin = torch.cat([ang.cos()] * 2, -1)[None, :, None], torch.cat([ang.sin()] * 2, -1)[None, :, None]
    rot = torch.cat([-x[..., d // 2:], x[..., : d // 2]], -1)
    return x * cos + rot * sin


# ---------------------------------------------------------------- chatglm2 / 3 / glm4
def chatglm_ckpt(mqa=True):
    groups = 2 if mqa else NH
    cfg = dict(model_type="chatglm", num_layers=L, hidden_size=H, ffn_hidden_size=192, num_attention_heads=NH, kv_channels=D,
               multi_query_attention=mqa, multi_query_group_num=2, layernorm_epsilon=1e-5, rmsnorm=True, add_qkv_bias=True,
               padded_vocab_size=V, seq_length=256, rope_ratio=1.0)
    sd = {"transformer.embedding.word_embeddings.weight": _w(V, H, s=0.5), "transformer.encoder.final_layernorm.weight": 1 + _w(H),
          "transformer.output_layer.weight": _w(V, H)}
    for i in range(L):
        p = f"transformer.encoder.layers.{i}."
        sd[p + "input_layernorm.weight"] = 1 + _w(H)
        sd[p + "post_attention_layernorm.weight"] = 1 + _w(H)
        sd[p + "self_attention.query_key_value.weight"] = _w((NH + 2 * groups) * D, H)
        sd[p + "self_attention.query_key_value.bias"] = _w((NH + 2 * groups) * D)
        sd[p + "self_attention.dense.weight"] = _w(H, NH * D)
        sd[p + "mlp.dense_h_to_4h.weight"] = _w(2 * 192, H)
        sd[p + "mlp.dense_4h_to_h.weight"] = _w(H, 192)
    return cfg, sd


def chatglm_forward(cfg, sd, ids):
    mqa, groups = cfg["multi_query_attention"], cfg["multi_query_group_num"]
    b, s = ids.shape
    rot = D // 2
    theta = 1.0 / (10000.0 ** (torch.arange(0, rot, 2).float() / rot))
    idx = torch.outer(torch.arange(s).float(), theta)
    cache = torch.stack([idx.cos(), idx.sin()], -1)                     # [s, rot/2, 2]

    def rope(x):                                                        # x [b, s, h, d]: pairs (2i, 2i+1) of the first half
        xr, xp = x[..., :rot], x[..., rot:]
        xs = xr.reshape(b, s, x.shape[2], rot // 2, 2)
        c = cache.view(1, s, 1, rot // 2, 2)
        o = torch.stack([xs[..., 0] * c[..., 0] - xs[..., 1] * c[..., 1], xs[..., 1] * c[..., 0] + xs[..., 0] * c[..., 1]], -1)
        return torch.cat([o.flatten(3), xp], -1)

    h = sd["transformer.embedding.word_embeddings.weight"][ids]
    for i in range(L):
        p = f"transformer.encoder.layers.{i}."
        x = _norm(h, sd[p + "input_layernorm.weight"], 1e-5)
        mixed = x @ sd[p + "self_attention.query_key_value.weight"].t() + sd[p + "self_attention.query_key_value.bias"]
        if mqa:
            q, k, v = mixed.split([NH * D, groups * D, groups * D], -1)
            q, k, v = q.view(b, s, NH, D), k.view(b, s, groups, D), v.view(b, s, groups, D)
        else:
            q, k, v = mixed.view(b, s, NH, 3 * D).split(D, -1)
        a = _attend(rope(q), rope(k), v)
        h = h + a @ sd[p + "self_attention.dense.weight"].t()
        x = _norm(h, sd[p + "post_attention_layernorm.weight"], 1e-5)
        g, u = (x @ sd[p + "mlp.dense_h_to_4h.weight"].t()).chunk(2, -1)
        h = h + (F.silu(g) * u) @ sd[p + "mlp.dense_4h_to_h.weight"].t()
    return _norm(h, sd["transformer.encoder.final_layernorm.weight"], 1e-5) @ sd["transformer.output_layer.weight"].t()


# ---------------------------------------------------------------- internlm2
def internlm2_ckpt():
    hkv = 2
    cfg = dict(model_type="internlm2", num_hidden_layers=L, hidden_size=H, intermediate_size=192, num_attention_heads=NH,
               num_key_value_heads=hkv, rms_norm_eps=1e-5, vocab_size=V, max_position_embeddings=256, rope_theta=10000.0)
    sd = {"model.tok_embeddings.weight": _w(V, H, s=0.5), "model.norm.weight": 1 + _w(H), "output.weight": _w(V, H)}
    for i in range(L):
        p = f"model.layers.{i}."
        sd[p + "attention.wqkv.weight"] = _w((NH + 2 * hkv) * D, H)
        sd[p + "attention.wo.weight"] = _w(H, NH * D)
        sd[p + "attention_norm.weight"], sd[p + "ffn_norm.weight"] = 1 + _w(H), 1 + _w(H)
        sd[p + "feed_forward.w1.weight"], sd[p + "feed_forward.w3.weight"] = _w(192, H), _w(192, H)
        sd[p + "feed_forward.w2.weight"] = _w(H, 192)
    return cfg, sd


def internlm2_forward(cfg, sd, ids):
    hkv, g = cfg["num_key_value_heads"], NH // cfg["num_key_value_heads"]
    b, s = ids.shape
    h = sd["model.tok_embeddings.weight"][ids]
    for i in range(L):
        p = f"model.layers.{i}."
        x = _norm(h, sd[p + "attention_norm.weight"], 1e-5)
        qkv = (x @ sd[p + "attention.wqkv.weight"].t()).view(b, s, hkv, g + 2, D)     # "b q (h gs d) -> b q h gs d"
        q = qkv[..., :g, :].reshape(b, s, NH, D)
        k, v = qkv[..., -2, :], qkv[..., -1, :]
        h = h + _attend(_rope_half(q), _rope_half(k), v) @ sd[p + "attention.wo.weight"].t()
        x = _norm(h, sd[p + "ffn_norm.weight"], 1e-5)
        h = h + (F.silu(x @ sd[p + "feed_forward.w1.weight"].t()) * (x @ sd[p + "feed_forward.w3.weight"].t())) \
            @ sd[p + "feed_forward.w2.weight"].t()
    return _norm(h, sd["model.norm.weight"], 1e-5) @ sd["output.weight"].t()


# ---------------------------------------------------------------- qwen (v1)
def qwen_ckpt():
    cfg = dict(model_type="qwen", num_hidden_layers=L, hidden_size=H, intermediate_size=2 * 192, num_attention_heads=NH,
               kv_channels=D, layer_norm_epsilon=1e-6, vocab_size=V, seq_length=256, rotary_emb_base=10000)
    sd = {"transformer.wte.weight": _w(V, H, s=0.5), "transformer.ln_f.weight": 1 + _w(H), "lm_head.weight": _w(V, H)}
    for i in range(L):
        p = f"transformer.h.{i}."
        sd[p + "ln_1.weight"], sd[p + "ln_2.weight"] = 1 + _w(H), 1 + _w(H)
        sd[p + "attn.c_attn.weight"], sd[p + "attn.c_attn.bias"] = _w(3 * H, H), _w(3 * H)
        sd[p + "attn.c_proj.weight"] = _w(H, H)
        sd[p + "mlp.w1.weight"], sd[p + "mlp.w2.weight"], sd[p + "mlp.c_proj.weight"] = _w(192, H), _w(192, H), _w(H, 192)
    return cfg, sd


def qwen_forward(cfg, sd, ids):
    b, s = ids.shape
    h = sd["transformer.wte.weight"][ids]
    for i in range(L):
        p = f"transformer.h.{i}."
        x = _norm(h, sd[p + "ln_1.weight"], 1e-6)
        q, k, v = (x @ sd[p + "attn.c_attn.weight"].t() + sd[p + "attn.c_attn.bias"]).split(H, dim=2)
        q, k, v = (t.view(b, s, NH, D) for t in (q, k, v))
        h = h + _attend(_rope_half(q), _rope_half(k), v) @ sd[p + "attn.c_proj.weight"].t()
        x = _norm(h, sd[p + "ln_2.weight"], 1e-6)
        a1, a2 = x @ sd[p + "mlp.w1.weight"].t(), x @ sd[p + "mlp.w2.weight"].t()
        h = h + (a1 * F.silu(a2)) @ sd[p + "mlp.c_proj.weight"].t()
    return _norm(h, sd["transformer.ln_f.weight"], 1e-6) @ sd["lm_head.weight"].t()


# ---------------------------------------------------------------- llama-named families (baichuan2, internlm, minicpm)
def llama_named_ckpt(kind, alibi=False):
    cfg = dict(model_type=kind, num_hidden_layers=L, hidden_size=H, intermediate_size=192, num_attention_heads=NH,
               rms_norm_eps=1e-6, vocab_size=V, max_position_embeddings=256)
    sd = {"model.embed_tokens.weight": _w(V, H, s=0.5), "model.norm.weight": 1 + _w(H), "lm_head.weight": _w(V, H)}
    for i in range(L):
        p = f"model.layers.{i}."
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = 1 + _w(H), 1 + _w(H)
        if kind == "baichuan":
            sd[p + "self_attn.W_pack.weight"] = _w(3 * H, H)
        else:
            for n in "qkv":
                sd[p + f"self_attn.{n}_proj.weight"] = _w(H, H)
                if kind == "internlm":
                    sd[p + f"self_attn.{n}_proj.bias"] = _w(H)
        sd[p + "self_attn.o_proj.weight"] = _w(H, H)
        if kind == "internlm":
            sd[p + "self_attn.o_proj.bias"] = _w(H)
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"] = _w(192, H), _w(192, H), _w(H, 192)
    if kind == "baichuan" and alibi:
        del cfg["max_position_embeddings"]   # the 13B configs carry model_max_length instead
        cfg["model_max_length"] = 256
    if kind == "baichuan":
        cfg["vocab_size"] = 125696          # Baichuan2: NormHead
        sd["model.embed_tokens.weight"], sd["lm_head.weight"] = _w(125696, H, s=0.5), _w(125696, H)
    if kind == "minicpm":
        cfg.update(scale_emb=12, scale_depth=1.4, dim_model_base=64)
        del sd["lm_head.weight"]
    return cfg, sd


def llama_named_forward(cfg, sd, ids):
    kind = cfg["model_type"]
    b, s = ids.shape
    h = sd["model.embed_tokens.weight"][ids]
    rs = 1.0
    if kind == "minicpm":
        h = h * cfg["scale_emb"]
        rs = cfg["scale_depth"] / math.sqrt(L)
    for i in range(L):
        p = f"model.layers.{i}."
        x = _norm(h, sd[p + "input_layernorm.weight"], 1e-6)
        if kind == "baichuan":
            proj = (x @ sd[p + "self_attn.W_pack.weight"].t()).unflatten(-1, (3, H))
            q, k, v = proj[..., 0, :], proj[..., 1, :], proj[..., 2, :]
        else:
            q, k, v = (x @ sd[p + f"self_attn.{n}_proj.weight"].t() + sd.get(p + f"self_attn.{n}_proj.bias", 0) for n in "qkv")
        q, k, v = (t.reshape(b, s, NH, D) for t in (q, k, v))
        if "max_position_embeddings" not in cfg:
            # Baichuan-13B: no rotary; bias = slope_h * key position (its _gen_alibi_mask), slopes 2^(-8 (h+1) / NH)
            slopes = torch.tensor([2.0 ** (-8.0 * (j + 1) / NH) for j in range(NH)])
            att = _attend(q, k, v, bias=slopes[:, None, None] * torch.arange(s)[None, None, :].float())
        else:
            att = _attend(_rope_half(q), _rope_half(k), v)
        o = att @ sd[p + "self_attn.o_proj.weight"].t() + sd.get(p + "self_attn.o_proj.bias", 0)
        h = h + o * rs
        x = _norm(h, sd[p + "post_attention_layernorm.weight"], 1e-6)
        m = (F.silu(x @ sd[p + "mlp.gate_proj.weight"].t()) * (x @ sd[p + "mlp.up_proj.weight"].t())) @ sd[p + "mlp.down_proj.weight"].t()
        h = h + m * rs
    h = _norm(h, sd["model.norm.weight"], 1e-6)
    if kind == "minicpm":
        return (h / (H / cfg["dim_model_base"])) @ sd["model.embed_tokens.weight"].t()
    if kind == "baichuan":
        return h @ F.normalize(sd["lm_head.weight"], dim=-1).t()
    return h @ sd["lm_head.weight"].t()


# ---------------------------------------------------------------- DeciLM: kv heads differ per layer
def decilm_ckpt():
    per = [4, 1][:L] if L == 2 else [4, 2, 1, 2][:L]
    cfg = dict(model_type="deci", num_hidden_layers=L, hidden_size=H, intermediate_size=192, num_attention_heads=NH,
               num_key_value_heads_per_layer=per, rms_norm_eps=1e-6, vocab_size=V, max_position_embeddings=256)
    sd = {"model.embed_tokens.weight": _w(V, H, s=0.5), "model.norm.weight": 1 + _w(H), "lm_head.weight": _w(V, H)}
    for i in range(L):
        p = f"model.layers.{i}."
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = 1 + _w(H), 1 + _w(H)
        sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.o_proj.weight"] = _w(H, H), _w(H, H)
        sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = _w(per[i] * D, H), _w(per[i] * D, H)
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"] = _w(192, H), _w(192, H), _w(H, 192)
    return cfg, sd


def decilm_forward(cfg, sd, ids):
    b, s = ids.shape
    h = sd["model.embed_tokens.weight"][ids]
    for i, nkv in enumerate(cfg["num_key_value_heads_per_layer"]):
        p = f"model.layers.{i}."
        x = _norm(h, sd[p + "input_layernorm.weight"], 1e-6)
        q = (x @ sd[p + "self_attn.q_proj.weight"].t()).reshape(b, s, NH, D)
        k = (x @ sd[p + "self_attn.k_proj.weight"].t()).reshape(b, s, nkv, D)
        v = (x @ sd[p + "self_attn.v_proj.weight"].t()).reshape(b, s, nkv, D)
        h = h + _attend(_rope_half(q), _rope_half(k), v) @ sd[p + "self_attn.o_proj.weight"].t()
        x = _norm(h, sd[p + "post_attention_layernorm.weight"], 1e-6)
        h = h + (F.silu(x @ sd[p + "mlp.gate_proj.weight"].t()) * (x @ sd[p + "mlp.up_proj.weight"].t())) @ sd[p + "mlp.down_proj.weight"].t()
    return _norm(h, sd["model.norm.weight"], 1e-6) @ sd["lm_head.weight"].t()


CASES = {
    "decilm_variable_gqa": (decilm_ckpt, decilm_forward),
    "chatglm3": (lambda: chatglm_ckpt(True), chatglm_forward),
    "chatglm_no_mqa": (lambda: chatglm_ckpt(False), chatglm_forward),
    "internlm2": (internlm2_ckpt, internlm2_forward),
    "qwen": (qwen_ckpt, qwen_forward),
    "baichuan2": (lambda: llama_named_ckpt("baichuan"), llama_named_forward),
    "baichuan2_13b": (lambda: llama_named_ckpt("baichuan", alibi=True), llama_named_forward),
    "internlm": (lambda: llama_named_ckpt("internlm"), llama_named_forward),
    "minicpm": (lambda: llama_named_ckpt("minicpm"), llama_named_forward),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_weight_map_reproduces_family_forward(case):
    torch.manual_seed(0)
    make, fwd = CASES[case]
    cfg, sd = make()
    ids = torch.randint(0, V, (2, 11))
    want = fwd(cfg, sd, ids)
    dc, canon = arch.ARCH_CONVERTERS[cfg["model_type"]](cfg, sd)
    got, past = arch.reference_forward(dc, canon, ids)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
    # incremental decoding over the returned KV equals the full forward
    step, _ = arch.reference_forward(dc, canon, ids[:, -1:], [(k[:, :, :-1], v[:, :, :-1]) for k, v in past])
    torch.testing.assert_close(step[:, 0], want[:, -1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", ["chatglm3", "qwen", "minicpm"])
def test_quantized_cpu_model_and_checkpoint_round_trip(case, tmp_path):
    from safetensors.torch import save_file
    from ipex_llm.transformers import AutoModelForCausalLM
    torch.manual_seed(1)
    make, fwd = CASES[case]
    cfg, sd = make()
    d = tmp_path / "ckpt"
    d.mkdir()
    json.dump(cfg, open(d / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    ids = torch.randint(0, V, (1, 9))
    want = fwd(cfg, sd, ids)
    model = AutoModelForCausalLM.from_pretrained(str(d), load_in_low_bit="sym_int8", trust_remote_code=True)
    assert isinstance(model, arch.NativeCausalLM)
    got, _ = model(ids)
    assert ((got - want).abs().mean() / want.abs().mean()).item() < 0.05
    out = model.generate(ids, max_new_tokens=4)
    assert out.shape == (1, 13)
    model.save_low_bit(str(tmp_path / "lb"))
    again = AutoModelForCausalLM.load_low_bit(str(tmp_path / "lb"))
    assert torch.equal(again.generate(ids, max_new_tokens=4), out)
    with pytest.raises(Exception):
        AutoModelForCausalLM.from_pretrained(str(tmp_path / "lb"))      # "Please use load_low_bit"


def test_longrope_factors_and_unsupported_rope_types():
    """longrope: short factors + attention factor below the original context, long factors above it, both equal to HF's
    ROPE_INIT_FUNCTIONS; rope types without an engine implementation are refused instead of silently unscaled."""
    import math
    import pytest
    from bigdl_amd.models.decoder import DecoderConfig, compute_inv_freq
    rot = 32
    short = [1.0 + 0.01 * i for i in range(rot // 2)]
    long_ = [2.0 + 0.05 * i for i in range(rot // 2)]
    rs = dict(rope_type="longrope", short_factor=short, long_factor=long_, original_max_position_embeddings=128)
    cfg = DecoderConfig(vocab_size=32, hidden_size=64, intermediate_size=64, num_layers=1, num_heads=2, num_kv_heads=2,
                        head_dim=rot, max_position=1024, rope_scaling=rs)
    base = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    want_factor = math.sqrt(1 + math.log(1024 / 128) / math.log(128))
    inv_s, f_s = compute_inv_freq(cfg, seq_len=100)
    inv_l, f_l = compute_inv_freq(cfg, seq_len=500)
    torch.testing.assert_close(inv_s, base / torch.tensor(short))
    torch.testing.assert_close(inv_l, base / torch.tensor(long_))
    assert abs(f_s - want_factor) < 1e-6 and abs(f_l - want_factor) < 1e-6
    try:
        from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
        from transformers import Phi3Config
        hf = Phi3Config(hidden_size=64, num_attention_heads=2, max_position_embeddings=1024,
                        original_max_position_embeddings=128, pad_token_id=0,
                        rope_scaling=dict(type="longrope", short_factor=short, long_factor=long_))
        inv_hf, f_hf = ROPE_INIT_FUNCTIONS["longrope"](hf, "cpu", seq_len=500)
        torch.testing.assert_close(inv_l, inv_hf)
        assert abs(f_hf - f_l) < 1e-6
    except (ImportError, KeyError, TypeError):
        pass
    cfg.rope_scaling = dict(rope_type="dynamic", factor=2.0)
    with pytest.raises(NotImplementedError):
        compute_inv_freq(cfg)
