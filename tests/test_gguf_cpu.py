"""GGUF reader / writer / loaders.  Fixtures are written by this package's own writer from random tiny HF models
(the reference's gguf tests download real files, which is not possible offline; parity against llama.cpp-written
files is therefore unpinned, the on-disk structures follow the GGUF v3 specification)."""
import pytest
import torch
from transformers import BloomConfig, BloomForCausalLM, LlamaConfig, LlamaForCausalLM, MixtralConfig, MixtralForCausalLM

from bigdl_amd.transformers.gguf import GGUFFileLoader, GGUFWriter
from bigdl_amd.transformers.gguf import gguf as G


def _permute(w, n_head):
    """HF -> llama.cpp row order (inverse of the loader's un-permute)."""
    rows = w.shape[0]
    return w.reshape(n_head, 2, rows // n_head // 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)


def _vocab_kv(wr, n):
    toks = ["<unk>", "<s>", "</s>"] + [f"▁t{i}" for i in range(n - 3)]
    wr.add("tokenizer.ggml.model", "llama")
    wr.add("tokenizer.ggml.tokens", toks)
    wr.add("tokenizer.ggml.scores", [0.0] * 3 + [-float(i) for i in range(n - 3)])
    wr.add("tokenizer.ggml.token_type", [2, 3, 3] + [1] * (n - 3))
    wr.add("tokenizer.ggml.bos_token_id", 1)
    wr.add("tokenizer.ggml.eos_token_id", 2)
    wr.add("tokenizer.ggml.unknown_token_id", 0)


def _write_llama(model, path, wtype, name="tiny llama", experts=0):
    cfg = model.config
    wr = GGUFWriter()
    wr.add("general.architecture", "llama")
    wr.add("general.name", name)
    wr.add("general.file_type", {G.GGML_F16: 1, G.GGML_Q4_0: 2, G.GGML_Q8_0: 7, G.GGML_Q4_K: 15, G.GGML_Q6_K: 18}[wtype])
    for k, v in (("context_length", cfg.max_position_embeddings), ("embedding_length", cfg.hidden_size),
                 ("block_count", cfg.num_hidden_layers), ("feed_forward_length", cfg.intermediate_size),
                 ("attention.head_count", cfg.num_attention_heads), ("attention.head_count_kv", cfg.num_key_value_heads)):
        wr.add("llama." + k, int(v))
    wr.add("llama.attention.layer_norm_rms_epsilon", float(cfg.rms_norm_eps))
    wr.add("llama.rope.freq_base", 10000.0)
    if experts:
        wr.add("llama.expert_count", experts)
        wr.add("llama.expert_used_count", cfg.num_experts_per_tok)
    _vocab_kv(wr, cfg.vocab_size)
    sd = model.state_dict()
    wr.add_tensor("token_embd.weight", sd["model.embed_tokens.weight"], wtype)
    wr.add_tensor("output_norm.weight", sd["model.norm.weight"], G.GGML_F32)
    wr.add_tensor("output.weight", sd["lm_head.weight"], wtype)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        wr.add_tensor(f"blk.{i}.attn_norm.weight", sd[p + "input_layernorm.weight"], G.GGML_F32)
        wr.add_tensor(f"blk.{i}.ffn_norm.weight", sd[p + "post_attention_layernorm.weight"], G.GGML_F32)
        wr.add_tensor(f"blk.{i}.attn_q.weight", _permute(sd[p + "self_attn.q_proj.weight"], cfg.num_attention_heads), wtype)
        wr.add_tensor(f"blk.{i}.attn_k.weight", _permute(sd[p + "self_attn.k_proj.weight"], cfg.num_key_value_heads), wtype)
        wr.add_tensor(f"blk.{i}.attn_v.weight", sd[p + "self_attn.v_proj.weight"], wtype)
        wr.add_tensor(f"blk.{i}.attn_output.weight", sd[p + "self_attn.o_proj.weight"], wtype)
        if experts:
            wr.add_tensor(f"blk.{i}.ffn_gate_inp.weight", sd[p + "mlp.gate.weight"], G.GGML_F32)
            gu = sd[p + "mlp.experts.gate_up_proj"]
            inter = gu.shape[1] // 2
            wr.add_tensor(f"blk.{i}.ffn_gate_exps.weight", gu[:, :inter].contiguous(), wtype)
            wr.add_tensor(f"blk.{i}.ffn_up_exps.weight", gu[:, inter:].contiguous(), wtype)
            wr.add_tensor(f"blk.{i}.ffn_down_exps.weight", sd[p + "mlp.experts.down_proj"], wtype)
        else:
            for a, b in (("ffn_gate", "mlp.gate_proj"), ("ffn_up", "mlp.up_proj"), ("ffn_down", "mlp.down_proj")):
                wr.add_tensor(f"blk.{i}.{a}.weight", sd[p + b + ".weight"], wtype)
    wr.write(path)


def _tiny_llama():
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-5, tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).eval()


def test_reader_roundtrips_metadata_and_all_tensor_types(tmp_path):
    torch.manual_seed(0)
    wr = GGUFWriter()
    wr.add("general.architecture", "llama")
    wr.add("some.int", 7)
    wr.add("some.float", 0.5)
    wr.add("some.flag", True)
    wr.add("some.strings", ["a", "bc", "déf"])
    wr.add("some.ints", [1, 2, 3])
    w = torch.randn(8, 256)
    types = [G.GGML_F32, G.GGML_F16, G.GGML_Q4_0, G.GGML_Q4_1, G.GGML_Q5_0, G.GGML_Q5_1, G.GGML_Q8_0, G.GGML_Q2_K,
             G.GGML_Q4_K, G.GGML_Q5_K, G.GGML_Q6_K]
    for t in types:
        wr.add_tensor(f"w.{t}", w, t)
    path = str(tmp_path / "t.gguf")
    wr.write(path)
    ld = GGUFFileLoader(path)
    assert ld.reader.version == 3
    assert ld.config["some.int"] == 7 and ld.config["some.float"] == 0.5 and ld.config["some.flag"] is True
    assert ld.config["some.strings"] == ["a", "bc", "déf"] and ld.config["some.ints"] == [1, 2, 3]
    tol = {G.GGML_F32: 0, G.GGML_F16: 2e-3, G.GGML_Q8_0: 0.02, G.GGML_Q6_K: 0.05, G.GGML_Q5_K: 0.08, G.GGML_Q5_0: 0.1,
           G.GGML_Q5_1: 0.1, G.GGML_Q4_0: 0.2, G.GGML_Q4_1: 0.2, G.GGML_Q4_K: 0.15, G.GGML_Q2_K: 0.6}
    for t in types:
        got = ld.tensor_loader.load(f"w.{t}")
        assert got.shape == w.shape
        assert ((got - w).abs().mean() / w.abs().mean()).item() <= tol[t], G.GGML_NAMES[t]
    raw, qt, shape = ld.tensor_loader.quantized(f"w.{G.GGML_Q4_0}")
    assert raw.numel() == 8 * 256 // 32 * 18 and shape == (8, 256)


def test_q4_0_block_layout_matches_specification(tmp_path):
    """q4_0: [fp16 d][16 bytes: low nibble = element i, high nibble = element i + 16], value (q - 8) * d."""
    import numpy as np
    d = np.float16(0.25)
    codes = np.arange(32) % 16
    blk = np.frombuffer(d.tobytes(), np.uint8).tolist() + [(int(codes[i]) | (int(codes[i + 16]) << 4)) for i in range(16)]
    wr = GGUFWriter()
    wr.add("general.architecture", "llama")
    wr.tensors.append(("x", (1, 32), G.GGML_Q4_0, bytes(blk)))
    path = str(tmp_path / "b.gguf")
    wr.write(path)
    got = GGUFFileLoader(path).tensor_loader.load("x")
    assert torch.equal(got[0], torch.tensor((codes - 8) * 0.25, dtype=torch.float32))


@pytest.mark.parametrize("wtype,tol", [(G.GGML_F16, 2e-3), (G.GGML_Q8_0, 0.03), (G.GGML_Q4_K, 0.3)])
def test_load_llama_gguf(tmp_path, wtype, tol):
    from ipex_llm.transformers import AutoModelForCausalLM
    m = _tiny_llama()
    path = str(tmp_path / "llama.gguf")
    _write_llama(m, path, wtype)
    ids = torch.randint(3, 320, (1, 9))
    with torch.no_grad():
        ref = m(ids).logits
    from bigdl_amd.transformers.gguf import load_gguf_model
    hf, tok = load_gguf_model(path, dtype=torch.float32)
    with torch.no_grad():
        got = hf(ids).logits
    assert ((got - ref).abs().mean() / ref.abs().mean()).item() < tol
    assert tok is not None and tok.bos_token == "<s>" and len(tok) >= 320
    # the public entry point: gguf -> low-bit model
    q, tok2 = AutoModelForCausalLM.from_gguf(path, low_bit="sym_int8")
    with torch.no_grad():
        got2 = q(ids).logits.float()
    assert ((got2 - ref).abs().mean() / ref.abs().mean()).item() < tol + 0.03
    assert type(q.model.layers[0].mlp.down_proj).__name__ == "LowBitLinear"


def test_load_mixtral_gguf(tmp_path):
    torch.manual_seed(0)
    cfg = MixtralConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, num_local_experts=4, num_experts_per_tok=2, max_position_embeddings=256,
                        rms_norm_eps=1e-5, sliding_window=None)
    m = MixtralForCausalLM(cfg).eval()
    path = str(tmp_path / "mixtral.gguf")
    _write_llama(m, path, G.GGML_F16, name="tiny Mixtral 4x", experts=4)
    from bigdl_amd.transformers.gguf import load_gguf_model
    hf, _ = load_gguf_model(path, dtype=torch.float32)
    assert type(hf).__name__ == "MixtralForCausalLM"
    ids = torch.randint(3, 320, (1, 7))
    with torch.no_grad():
        assert ((hf(ids).logits - m(ids).logits).abs().mean() / m(ids).logits.abs().mean()).item() < 5e-3


def test_load_bloom_gguf(tmp_path):
    torch.manual_seed(0)
    cfg = BloomConfig(vocab_size=320, hidden_size=128, n_layer=2, n_head=4)
    m = BloomForCausalLM(cfg).eval()
    sd = m.state_dict()
    wr = GGUFWriter()
    wr.add("general.architecture", "bloom")
    wr.add("general.name", "tiny bloom")
    wr.add("general.file_type", 1)
    for k, v in (("embedding_length", 128), ("block_count", 2), ("attention.head_count", 4)):
        wr.add("bloom." + k, v)
    wr.add("bloom.attention.layer_norm_epsilon", 1e-5)
    _vocab_kv(wr, 320)
    top = {"token_embd": "transformer.word_embeddings", "token_embd_norm": "transformer.word_embeddings_layernorm",
           "output_norm": "transformer.ln_f", "output": "lm_head"}
    for g, h in top.items():
        for kind in ("weight", "bias"):
            if f"{h}.{kind}" in sd:
                wr.add_tensor(f"{g}.{kind}", sd[f"{h}.{kind}"], G.GGML_F32)
    blk = {"attn_norm": "input_layernorm", "attn_qkv": "self_attention.query_key_value", "attn_output": "self_attention.dense",
           "ffn_norm": "post_attention_layernorm", "ffn_up": "mlp.dense_h_to_4h", "ffn_down": "mlp.dense_4h_to_h"}
    for i in range(2):
        for g, h in blk.items():
            for kind in ("weight", "bias"):
                t = sd[f"transformer.h.{i}.{h}.{kind}"]
                if g == "attn_qkv":        # HF per-head [q, k, v] -> gguf [all q | all k | all v]
                    t = t.reshape(4, 3, 32, *t.shape[1:]).transpose(0, 1).reshape(t.shape)
                wr.add_tensor(f"blk.{i}.{g}.{kind}", t, G.GGML_F32)
    path = str(tmp_path / "bloom.gguf")
    wr.write(path)
    from bigdl_amd.transformers.gguf import load_gguf_model
    hf, _ = load_gguf_model(path, dtype=torch.float32)
    ids = torch.randint(3, 320, (1, 7))
    with torch.no_grad():
        torch.testing.assert_close(hf(ids).logits, m(ids).logits, rtol=1e-4, atol=1e-4)


def _save_tiny_checkpoint(tmp_path):
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from transformers import PreTrainedTokenizerFast
    m = _tiny_llama()
    d = tmp_path / "hf"
    m.save_pretrained(str(d))
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    vocab.update({f"w{i}": i + 3 for i in range(317)})
    t = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    t.pre_tokenizer = Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=t, bos_token="<s>", eos_token="</s>", unk_token="<unk>").save_pretrained(str(d))
    return m, str(d)


def test_llm_convert_and_native_wrappers(tmp_path):
    from ipex_llm import llm_convert
    from ipex_llm.transformers import BigdlNativeForCausalLM
    m, d = _save_tiny_checkpoint(tmp_path)
    out = llm_convert(model=d, outfile=str(tmp_path / "out"), model_family="llama", outtype="int8", model_format="pth")
    assert out.endswith(".gguf")
    ld = GGUFFileLoader(out)
    assert ld.config["general.file_type"] == 7 and ld.tensor_loader.info("blk.0.attn_q.weight")["type"] == G.GGML_Q8_0
    llm = BigdlNativeForCausalLM.from_pretrained(out, model_family="llama", device="cpu")
    ids = torch.randint(3, 320, (1, 8))
    with torch.no_grad():
        ref = m(ids).logits
        got = llm.model(ids).logits.float()
    assert ((got - ref).abs().mean() / ref.abs().mean()).item() < 0.05
    res = llm("w1 w2 w3", max_tokens=5, temperature=0.0)
    assert res["usage"]["completion_tokens"] <= 5 and isinstance(res["choices"][0]["text"], str)
    assert len(llm.embed("w4 w5")) == 256
    # streaming yields the same text piece by piece; chat / embedding responses follow the OpenAI shapes
    pieces = list(llm("w1 w2 w3", max_tokens=5, temperature=0.0, stream=True))
    assert "".join(p["choices"][0]["text"] for p in pieces) == res["choices"][0]["text"]
    assert pieces[-1]["choices"][0]["finish_reason"] in ("stop", "length")
    chat = llm.create_chat_completion([{"role": "user", "content": "w7 w8"}], max_tokens=3, temperature=0.0)
    assert chat["object"] == "chat.completion" and chat["choices"][0]["message"]["role"] == "assistant"
    emb = llm.create_embedding(["w1", "w2 w3"])
    assert [d["index"] for d in emb["data"]] == [0, 1] and len(emb["data"][1]["embedding"]) == 256
    toks = llm.tokenize("w1 w2")
    assert llm.eval(toks).shape == (len(toks), 320)
    from bigdl_amd.ggml.model import Bloom, LlamaCache
    with pytest.raises(ValueError, match="bloom"):
        Bloom(out)
    # the remaining llama-cpp-python accessors
    assert llm.n_embd() == 256 and llm.n_vocab() == 320 and llm.n_ctx() == 512
    assert llm.sample(toks, temp=0.0, repeat_penalty=1.0) == int(llm.eval(toks)[-1].argmax())
    assert 0 <= llm.sample(toks, top_k=5, top_p=0.9, temp=0.7) < 320
    lp = llm.logits_to_logprobs([0.0, 0.0])
    assert abs(lp[0] + 0.6931) < 1e-3 and llm.longest_token_prefix([1, 2, 3], [1, 2, 4]) == 2
    cache = LlamaCache()
    cache[[1, 2]] = "a"
    cache[[1, 2, 3]] = "b"
    assert cache[[1, 2, 3, 4]] == "b" and [1, 2, 9] in cache and [2] not in cache
    llm.set_cache(cache)
    llm.load_state(llm.save_state())
    llm.reset()
    # a completion cut short by a stop string leaves no generation running behind it: the worker is cancelled and
    # joined, so the next completion has the model (KV cache, decode graph) to itself and reproduces the first one
    import threading
    full = llm("w1 w2 w3", max_tokens=12, temperature=0.0)["choices"][0]["text"]
    words = full.split()
    if len(words) >= 3:
        before = threading.active_count()
        cut = llm("w1 w2 w3", max_tokens=12, temperature=0.0, stop=[words[1]])
        assert cut["choices"][0]["finish_reason"] == "stop" and words[1] not in cut["choices"][0]["text"]
        assert threading.active_count() <= before
        again = llm("w1 w2 w3", max_tokens=12, temperature=0.0)["choices"][0]["text"]
        assert again == full
    it = llm.generate(llm.tokenize("w1 w2 w3"), temp=0.0, max_tokens=64)
    next(it)
    it.close()                                      # abandoned iterator: GeneratorExit cancels the worker
    assert llm("w1 w2 w3", max_tokens=12, temperature=0.0)["choices"][0]["text"] == full


def test_gptq_checkpoint_loads_without_auto_gptq(tmp_path):
    """A GPTQ-packed checkpoint (qweight / qzeros / scales / g_idx + quantization_config) -> asym_int4 model."""
    import json
    from safetensors.torch import save_file
    from ipex_llm.transformers import AutoModelForCausalLM
    from bigdl_amd.transformers.gptq import pack_gptq
    m, d = _save_tiny_checkpoint(tmp_path)
    sd = {}
    deq = {}
    for k, v in m.state_dict().items():
        if v.dim() == 2 and "proj" in k:
            qw, qz, sc, gi = pack_gptq(v, group_size=64)
            base = k[:-len(".weight")]
            sd.update({base + ".qweight": qw, base + ".qzeros": qz, base + ".scales": sc, base + ".g_idx": gi})
        else:
            sd[k] = v.clone()
    gdir = tmp_path / "gptq"
    gdir.mkdir()
    save_file(sd, str(gdir / "model.safetensors"))
    cfg = json.load(open(d + "/config.json"))
    cfg["quantization_config"] = {"quant_method": "gptq", "bits": 4, "group_size": 64, "desc_act": False, "sym": False}
    json.dump(cfg, open(gdir / "config.json", "w"))
    q = AutoModelForCausalLM.from_pretrained(str(gdir), load_in_4bit=True)
    assert type(q.model.layers[0].mlp.down_proj).__name__ == "LowBitLinear"
    ids = torch.randint(3, 320, (1, 8))
    with torch.no_grad():
        ref = m(ids).logits
        got = q(ids).logits.float()
    assert ((got - ref).abs().mean() / ref.abs().mean()).item() < 0.25        # 4-bit RTN with groups of 64
