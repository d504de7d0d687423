"""GGUF file -> HuggingFace model + tokenizer.

Capability map to the reference python/llm/src/ipex_llm/transformers/gguf/api.py:31-75 (family dispatch on
``general.architecture`` / ``general.name``) and gguf/models/{llama,mistral,mixtral,baichuan,bloom,falcon,mpt,yuan2}.py
(config from metadata, tensor-name mapping, llama q/k un-permute, tokenizer rebuild).

The model is built from the stock HuggingFace classes on the meta device and filled tensor by tensor from the mapped
file (dequantized to ``dtype`` by the C++ host library); ``AutoModelForCausalLM.from_gguf`` then runs the normal
low-bit conversion.  The tokenizer is rebuilt from the gguf vocabulary with HuggingFace's own gguf tokenizer
converters.  ``yuan2`` is not part of transformers: it is served by this package's own model class (models/yuan.py).
"""
from __future__ import annotations

import re

import torch

from ...utils.common import invalidInputError
from .gguf import GGUFFileLoader

# general.file_type -> the closest low-bit format of this package (reference api.py:22-28)
qtype_map = {0: "fp16", 1: "fp16", 2: "sym_int4", 3: "asym_int4", 7: "sym_int8", 8: "sym_int5", 9: "asym_int5",
             10: "q2_k", 12: "q4_k", 14: "q4_k", 15: "q4_k", 16: "q5_k", 17: "q5_k", 18: "q6_k"}


def _unpermute(w: torch.Tensor, n_head: int) -> torch.Tensor:
    """llama.cpp stores q/k rows pair-interleaved per head (its RoPE convention); HF wants the two halves."""
    rows = w.shape[0]
    return w.reshape(n_head, rows // n_head // 2, 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)


def _fill(model, name: str, tensor: torch.Tensor, dtype):
    """Materialize one parameter of a meta-device model."""
    mod_name, _, pname = name.rpartition(".")
    mod = model.get_submodule(mod_name)
    old = getattr(mod, pname)
    invalidInputError(tuple(old.shape) == tuple(tensor.shape),
                      f"{name}: gguf tensor shape {tuple(tensor.shape)} != model shape {tuple(old.shape)}")
    t = tensor.to(dtype if tensor.dtype.is_floating_point else tensor.dtype).contiguous()
    if isinstance(old, torch.nn.Parameter):
        setattr(mod, pname, torch.nn.Parameter(t, requires_grad=False))
    else:
        mod.register_buffer(pname, t)


def _finish(model, dtype):
    """Buffers such as rotary inv_freq are created on the meta device with the module: rebuild them."""
    leftover = [n for n, p in list(model.named_parameters()) if p.device.type == "meta"]
    if hasattr(model, "lm_head") and "lm_head.weight" in leftover:
        emb = model.get_input_embeddings().weight
        model.lm_head.weight = torch.nn.Parameter(emb.data.clone(), requires_grad=False)     # tied output matrix
        leftover.remove("lm_head.weight")
    invalidInputError(not leftover, f"gguf file has no tensors for {leftover[:5]}")
    for mod in model.modules():
        for bname, buf in list(mod._buffers.items()):
            if buf is not None and buf.device.type == "meta":
                if bname in ("inv_freq", "original_inv_freq") and hasattr(mod, "config"):
                    mod._buffers[bname] = _rope_inv_freq(mod)
                else:
                    mod._buffers[bname] = torch.zeros(buf.shape, dtype=buf.dtype)
    model.eval()
    return model


def _rope_inv_freq(rotary):
    """Recompute a rotary module's frequencies on the host (its buffers were created on the meta device)."""
    fn = getattr(rotary, "rope_init_fn", None)
    if fn is None:
        rope_type = getattr(rotary, "rope_type", "default")
        if rope_type == "default" and hasattr(rotary, "compute_default_rope_parameters"):
            fn = rotary.compute_default_rope_parameters
        else:
            from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
            fn = ROPE_INIT_FUNCTIONS[rope_type]
    inv, scaling = fn(rotary.config, "cpu")
    rotary.attention_scaling = scaling
    return inv


def _tokenizer(loader: GGUFFileLoader, arch: str):
    from transformers import PreTrainedTokenizerFast
    from transformers.integrations.ggml import GGUF_TO_FAST_CONVERTERS, convert_gguf_tokenizer
    td = loader.tokenizer_dict()
    if not td.get("tokens"):
        return None
    key = arch if arch in GGUF_TO_FAST_CONVERTERS else ("gpt2" if td.get("tokenizer_type") == "gpt2" else "llama")
    try:
        from types import SimpleNamespace
        proto = SimpleNamespace(**{k: td.get(k) for k in ("tokenizer_type", "tokens", "scores", "token_type", "merges",
                                                          "bos_token_id", "eos_token_id", "unk_token_id", "pad_token_id",
                                                          "add_prefix_space")})
        fast, extra = convert_gguf_tokenizer(key, proto.__dict__)
    except Exception:      # converter signature differs between transformers versions: fall back to a plain vocab
        from tokenizers import Tokenizer
        from tokenizers.models import WordLevel
        fast, extra = Tokenizer(WordLevel({t: i for i, t in enumerate(td["tokens"])}, unk_token=td["tokens"][0])), {}
    toks = td["tokens"]
    kw = {}
    for name, key_ in (("bos_token", "bos_token_id"), ("eos_token", "eos_token_id"), ("unk_token", "unk_token_id"),
                       ("pad_token", "pad_token_id")):
        i = td.get(key_)
        if i is not None and 0 <= i < len(toks):
            kw[name] = toks[i]
    kw.update({k: v for k, v in (extra or {}).items() if k not in kw})
    return PreTrainedTokenizerFast(tokenizer_object=fast, **kw)


# ------------------------------------------------------------------------------------------------
# llama-layout families (llama, mistral, mixtral, qwen2, baichuan)
# ------------------------------------------------------------------------------------------------
_LLAMA_MAP = {"attn_norm": "input_layernorm", "attn_q": "self_attn.q_proj", "attn_k": "self_attn.k_proj",
              "attn_v": "self_attn.v_proj", "attn_output": "self_attn.o_proj", "ffn_norm": "post_attention_layernorm",
              "ffn_gate": "mlp.gate_proj", "ffn_up": "mlp.up_proj", "ffn_down": "mlp.down_proj"}


def get_llama_module_name(name: str):
    if name == "token_embd.weight":
        return "model.embed_tokens.weight"
    if name == "output_norm.weight":
        return "model.norm.weight"
    if name == "output.weight":
        return "lm_head.weight"
    m = re.match(r"blk\.(\d+)\.(\w+)\.(weight|bias)$", name)
    if m and m.group(2) in _LLAMA_MAP:
        return f"model.layers.{m.group(1)}.{_LLAMA_MAP[m.group(2)]}.{m.group(3)}"
    return None


def _llama_like_config(cfg, arch, cls, **extra):
    p = arch + "."
    n_head = cfg[p + "attention.head_count"]
    kw = dict(vocab_size=len(cfg["tokenizer.ggml.tokens"]) if "tokenizer.ggml.tokens" in cfg else cfg.get(p + "vocab_size"),
              hidden_size=cfg[p + "embedding_length"], intermediate_size=cfg[p + "feed_forward_length"],
              num_hidden_layers=cfg[p + "block_count"], num_attention_heads=n_head,
              num_key_value_heads=cfg.get(p + "attention.head_count_kv", n_head), hidden_act="silu",
              max_position_embeddings=cfg.get(p + "context_length", 4096),
              rms_norm_eps=cfg.get(p + "attention.layer_norm_rms_epsilon", 1e-5), use_cache=True,
              bos_token_id=cfg.get("tokenizer.ggml.bos_token_id"), eos_token_id=cfg.get("tokenizer.ggml.eos_token_id"),
              tie_word_embeddings=False)
    if p + "rope.freq_base" in cfg:
        kw["rope_theta"] = float(cfg[p + "rope.freq_base"])
    kw.update(extra)
    return cls(**kw)


def _load_llama_like(loader, dtype, config, model_cls, unpermute: bool, experts: int = 0):
    cfg = loader.config
    tl = loader.tensor_loader
    n_head, n_kv = config.num_attention_heads, config.num_key_value_heads
    with torch.device("meta"):
        model = model_cls(config)
    expert_parts = {}
    for name in tl.names():
        t = tl.load(name, torch.float32)
        m = re.match(r"blk\.(\d+)\.ffn_(gate|up|down)(?:\.(\d+)|_exps)\.weight$", name)
        if experts and m:                                         # per-expert or stacked expert tensors
            li, kind, e = int(m.group(1)), m.group(2), m.group(3)
            d = expert_parts.setdefault(li, {})
            if e is None:
                d[kind] = t
            else:
                d.setdefault(kind, {})[int(e)] = t
            continue
        if experts and re.match(r"blk\.(\d+)\.ffn_gate_inp\.weight$", name):
            li = int(name.split(".")[1])
            _fill(model, f"model.layers.{li}.mlp.gate.weight", t, dtype)
            continue
        hf = get_llama_module_name(name)
        if hf is None:
            continue                                              # rope_freqs etc.
        if unpermute and hf.endswith("q_proj.weight"):
            t = _unpermute(t, n_head)
        elif unpermute and hf.endswith("k_proj.weight"):
            t = _unpermute(t, n_kv)
        elif unpermute and hf.endswith("q_proj.bias"):
            t = _unpermute(t, n_head)
        elif unpermute and hf.endswith("k_proj.bias"):
            t = _unpermute(t, n_kv)
        _fill(model, hf, t, dtype)
    for li, d in expert_parts.items():
        def stack(kind):
            v = d[kind]
            return v if isinstance(v, torch.Tensor) else torch.stack([v[e] for e in range(experts)])
        gate_up = torch.cat([stack("gate"), stack("up")], dim=1)              # [E, 2I, H]
        _fill(model, f"model.layers.{li}.mlp.experts.gate_up_proj", gate_up, dtype)
        _fill(model, f"model.layers.{li}.mlp.experts.down_proj", stack("down"), dtype)
    return _finish(model, dtype)


def load_gguf_llama(loader, dtype=torch.float, low_bit="sym_int4"):
    from transformers import LlamaConfig, LlamaForCausalLM
    config = _llama_like_config(loader.config, "llama", LlamaConfig)
    return _load_llama_like(loader, dtype, config, LlamaForCausalLM, True), _tokenizer(loader, "llama")


def load_gguf_mistral(loader, dtype=torch.float, low_bit="sym_int4"):
    from transformers import MistralConfig, MistralForCausalLM
    config = _llama_like_config(loader.config, "llama", MistralConfig, sliding_window=None)
    return _load_llama_like(loader, dtype, config, MistralForCausalLM, True), _tokenizer(loader, "llama")


def load_gguf_mixtral(loader, dtype=torch.float, low_bit="sym_int4"):
    from transformers import MixtralConfig, MixtralForCausalLM
    cfg = loader.config
    config = _llama_like_config(cfg, "llama", MixtralConfig, num_local_experts=cfg["llama.expert_count"],
                                num_experts_per_tok=cfg["llama.expert_used_count"], sliding_window=None)
    return (_load_llama_like(loader, dtype, config, MixtralForCausalLM, True, experts=cfg["llama.expert_count"]),
            _tokenizer(loader, "llama"))


def load_gguf_qwen2(loader, dtype=torch.float, low_bit="sym_int4"):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    config = _llama_like_config(loader.config, "qwen2", Qwen2Config)
    return _load_llama_like(loader, dtype, config, Qwen2ForCausalLM, False), _tokenizer(loader, "qwen2")


def load_gguf_baichuan(loader, dtype=torch.float, low_bit="sym_int4"):
    """Baichuan-7B is the llama architecture with its own vocabulary; the 13B ALiBi variant needs the model's remote
    code and is rejected."""
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = loader.config
    invalidInputError(cfg.get("baichuan.rope.dimension_count") is not None,
                      "ALiBi Baichuan (13B) gguf files need the model's own code, which transformers does not ship")
    config = _llama_like_config(cfg, "baichuan", LlamaConfig)
    return _load_llama_like(loader, dtype, config, LlamaForCausalLM, True), _tokenizer(loader, "llama")


# ------------------------------------------------------------------------------------------------
# bloom / falcon / mpt
# ------------------------------------------------------------------------------------------------
def load_gguf_bloom(loader, dtype=torch.float, low_bit=None):
    from transformers import BloomConfig, BloomForCausalLM
    cfg, tl = loader.config, loader.tensor_loader
    n_head, hidden = cfg["bloom.attention.head_count"], cfg["bloom.embedding_length"]
    config = BloomConfig(vocab_size=len(cfg["tokenizer.ggml.tokens"]), hidden_size=hidden, n_layer=cfg["bloom.block_count"],
                         n_head=n_head, layer_norm_epsilon=cfg.get("bloom.attention.layer_norm_epsilon", 1e-5),
                         bos_token_id=cfg.get("tokenizer.ggml.bos_token_id"), eos_token_id=cfg.get("tokenizer.ggml.eos_token_id"))
    with torch.device("meta"):
        model = BloomForCausalLM(config)
    top = {"token_embd": "transformer.word_embeddings", "token_embd_norm": "transformer.word_embeddings_layernorm",
           "output_norm": "transformer.ln_f", "output": "lm_head"}
    blk = {"attn_norm": "input_layernorm", "attn_qkv": "self_attention.query_key_value",
           "attn_output": "self_attention.dense", "ffn_norm": "post_attention_layernorm", "ffn_up": "mlp.dense_h_to_4h",
           "ffn_down": "mlp.dense_4h_to_h"}
    hd = hidden // n_head
    for name in tl.names():
        t = tl.load(name, torch.float32)
        m = re.match(r"blk\.(\d+)\.(\w+)\.(weight|bias)$", name)
        if m and m.group(2) in blk:
            if m.group(2) == "attn_qkv":          # gguf: [q heads | k heads | v heads]; HF: per head [q, k, v]
                t = t.reshape(3, n_head, hd, *t.shape[1:]).transpose(0, 1).reshape(t.shape)
            _fill(model, f"transformer.h.{m.group(1)}.{blk[m.group(2)]}.{m.group(3)}", t, dtype)
            continue
        base, _, kind = name.rpartition(".")
        if base in top:
            _fill(model, f"{top[base]}.{kind}", t, dtype)
    return _finish(model, dtype), _tokenizer(loader, "bloom")


def load_gguf_falcon(loader, dtype=torch.float, low_bit=None):
    from transformers import FalconConfig, FalconForCausalLM
    cfg, tl = loader.config, loader.tensor_loader
    n_head = cfg["falcon.attention.head_count"]
    n_kv = cfg.get("falcon.attention.head_count_kv", 1)
    config = FalconConfig(vocab_size=len(cfg["tokenizer.ggml.tokens"]), hidden_size=cfg["falcon.embedding_length"],
                          num_hidden_layers=cfg["falcon.block_count"], num_attention_heads=n_head, num_kv_heads=n_kv,
                          multi_query=n_kv == 1, new_decoder_architecture=n_kv > 1, parallel_attn=True, bias=False,
                          layer_norm_epsilon=cfg.get("falcon.attention.layer_norm_epsilon", 1e-5),
                          bos_token_id=cfg.get("tokenizer.ggml.bos_token_id"), eos_token_id=cfg.get("tokenizer.ggml.eos_token_id"))
    with torch.device("meta"):
        model = FalconForCausalLM(config)
    top = {"token_embd": "transformer.word_embeddings", "output_norm": "transformer.ln_f", "output": "lm_head"}
    blk = {"attn_norm": "ln_attn" if n_kv > 1 else "input_layernorm", "attn_norm_2": "ln_mlp",
           "attn_qkv": "self_attention.query_key_value", "attn_output": "self_attention.dense",
           "ffn_up": "mlp.dense_h_to_4h", "ffn_down": "mlp.dense_4h_to_h"}
    for name in tl.names():
        t = tl.load(name, torch.float32)
        m = re.match(r"blk\.(\d+)\.(\w+)\.(weight|bias)$", name)
        if m and m.group(2) in blk:
            _fill(model, f"transformer.h.{m.group(1)}.{blk[m.group(2)]}.{m.group(3)}", t, dtype)
            continue
        base, _, kind = name.rpartition(".")
        if base in top:
            _fill(model, f"{top[base]}.{kind}", t, dtype)
    return _finish(model, dtype), _tokenizer(loader, "falcon")


def load_gguf_mpt(loader, dtype=torch.float, low_bit=None):
    from transformers import MptConfig, MptForCausalLM
    cfg, tl = loader.config, loader.tensor_loader
    config = MptConfig(vocab_size=len(cfg["tokenizer.ggml.tokens"]), d_model=cfg["mpt.embedding_length"],
                       n_layers=cfg["mpt.block_count"], n_heads=cfg["mpt.attention.head_count"],
                       max_seq_len=cfg.get("mpt.context_length", 2048), no_bias=True,
                       layer_norm_epsilon=cfg.get("mpt.attention.layer_norm_epsilon", 1e-5))
    with torch.device("meta"):
        model = MptForCausalLM(config)
    top = {"token_embd": "transformer.wte", "output_norm": "transformer.norm_f", "output": "lm_head"}
    blk = {"attn_norm": "norm_1", "attn_qkv": "attn.Wqkv", "attn_output": "attn.out_proj", "ffn_norm": "norm_2",
           "ffn_up": "ffn.up_proj", "ffn_down": "ffn.down_proj"}
    for name in tl.names():
        t = tl.load(name, torch.float32)
        m = re.match(r"blk\.(\d+)\.(\w+)\.(weight|bias)$", name)
        if m and m.group(2) in blk:
            target = f"transformer.blocks.{m.group(1)}.{blk[m.group(2)]}.{m.group(3)}"
        else:
            base, _, kind = name.rpartition(".")
            target = f"{top[base]}.{kind}" if base in top else None
        if target is None:
            continue
        try:
            _fill(model, target, t, dtype)
        except AttributeError:
            pass                                   # no_bias models have no bias parameters
    return _finish(model, dtype), _tokenizer(loader, "gpt2")


_YUAN_MAP = dict(_LLAMA_MAP, lf_output_norm="self_attn.lf_gate.output_layernorm", conv1="self_attn.lf_gate.conv1",
                 conv2="self_attn.lf_gate.conv2")


def load_gguf_yuan2(loader, dtype=torch.float, low_bit="sym_int4"):
    """Yuan 2.0 files identify themselves as architecture ``llama`` with "yuan" in ``general.name``; next to the llama
    tensors every block carries the localized filter (``conv1`` / ``conv2`` weight + bias, ``lf_output_norm``).  q / k
    rows are stored in model order (no llama.cpp head permutation).  The model class is this package's own
    (bigdl_amd/models/yuan.py); reference builder: transformers/gguf/models/yuan2.py:26-130."""
    from ...models.yuan import YuanConfig, YuanForCausalLM
    cfg = loader.config
    config = YuanConfig(vocab_size=len(cfg["tokenizer.ggml.tokens"]), hidden_size=cfg["llama.embedding_length"],
                        intermediate_size=cfg["llama.feed_forward_length"], num_hidden_layers=cfg["llama.block_count"],
                        num_attention_heads=cfg["llama.attention.head_count"],
                        max_position_embeddings=cfg["llama.context_length"],
                        rms_norm_eps=cfg["llama.attention.layer_norm_rms_epsilon"],
                        rope_theta=float(cfg.get("llama.rope.freq_base", 10000.0)),
                        pad_token_id=cfg.get("tokenizer.ggml.padding_token_id"), bos_token_id=cfg.get("tokenizer.ggml.bos_token_id"),
                        eos_token_id=cfg.get("tokenizer.ggml.eos_token_id"),
                        sep_token_id=cfg.get("tokenizer.ggml.seperator_token_id"))
    with torch.device("meta"):
        model = YuanForCausalLM(config)
    tl = loader.tensor_loader
    top = {"token_embd.weight": "model.embed_tokens.weight", "output_norm.weight": "model.norm.weight",
           "output.weight": "lm_head.weight"}
    for name in tl.names():
        hf = top.get(name)
        m = re.match(r"blk\.(\d+)\.(\w+)\.(weight|bias)$", name)
        if hf is None and m and m.group(2) in _YUAN_MAP:
            hf = f"model.layers.{m.group(1)}.{_YUAN_MAP[m.group(2)]}.{m.group(3)}"
        if hf is None:
            continue
        t = tl.load(name, torch.float32)
        if ".conv" in hf and hf.endswith("weight") and t.dim() != 4:
            out_c = t.shape[0]
            t = t.reshape(out_c, -1, 2, 1)                        # files that flatten the [out, in, 2, 1] kernel
        _fill(model, hf, t, dtype)
    model = _finish(model, dtype)
    # the rotary table is a buffer, not a tensor of the file
    for layer in model.model.layers:
        att = layer.self_attn
        att.inv_freq = 1.0 / (config.rope_theta ** (torch.arange(0, att.head_dim, 2).float() / att.head_dim))
    tok = _tokenizer(loader, "llama")
    if tok is not None:
        tok.add_tokens(["<sep>", "<pad>", "<mask>", "<predict>", "<FIM_SUFFIX>", "<FIM_PREFIX>", "<FIM_MIDDLE>",
                        "<commit_before>", "<commit_msg>", "<commit_after>", "<jupyter_start>", "<jupyter_text>",
                        "<jupyter_code>", "<jupyter_output>", "<empty_output>"], special_tokens=True)
    return model, tok


def load_gguf_model(fpath: str, dtype: torch.dtype = torch.float, low_bit: str = "sym_int4"):
    loader = GGUFFileLoader(fpath)
    family = loader.config["general.architecture"]
    ftype = loader.config.get("general.file_type")
    invalidInputError(ftype is None or ftype in qtype_map, f"Unsupported gguf quantize type: {ftype}")
    with torch.no_grad():
        if family == "llama":
            general_name = str(loader.config.get("general.name", "")).lower()
            if "mixtral" in general_name or loader.config.get("llama.expert_count", 0):
                return load_gguf_mixtral(loader, dtype, low_bit)
            if "mistral" in general_name:
                return load_gguf_mistral(loader, dtype, low_bit)
            if "yuan" in general_name:
                return load_gguf_yuan2(loader, dtype, low_bit)
            return load_gguf_llama(loader, dtype, low_bit)
        if family == "qwen2":
            return load_gguf_qwen2(loader, dtype, low_bit)
        if family == "baichuan":
            return load_gguf_baichuan(loader, dtype, low_bit)
        if family == "bloom":
            return load_gguf_bloom(loader, dtype)
        if family == "falcon":
            return load_gguf_falcon(loader, dtype)
        if family == "mpt":
            return load_gguf_mpt(loader, dtype)
    invalidInputError(False, f"Unsupported model family: {family}")
