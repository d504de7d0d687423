"""AutoModel / optimize_model API on the CPU with tiny random-init HF checkpoints (no network)."""
import json
import os
import struct

import pytest
import torch
import transformers
from transformers import LlamaConfig, LlamaForCausalLM

from bigdl_amd import qtypes as Q


@pytest.fixture(scope="module")
def tiny_llama(tmp_path_factory):
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256,
                      tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    d = str(tmp_path_factory.mktemp("tiny_llama"))
    m.save_pretrained(d)
    return d, m


def _rel(a, b):
    return ((a - b).abs().mean() / b.abs().mean()).item()


def test_ipex_llm_alias_imports():
    import ipex_llm
    from ipex_llm.transformers import AutoModelForCausalLM  # noqa: F401
    from ipex_llm.transformers.low_bit_linear import LowBitLinear  # noqa: F401
    from ipex_llm.ggml.quantize import ggml_tensor_qtype
    from ipex_llm.utils.common import invalidInputError  # noqa: F401
    import bigdl.llm
    assert ggml_tensor_qtype["sym_int4"] == 2 and ggml_tensor_qtype["nf4"] == 10 and ggml_tensor_qtype["fp8"] == 19
    assert ipex_llm.__version__ == bigdl.llm.__version__
    assert callable(ipex_llm.optimize_model)


@pytest.mark.parametrize("low_bit,tol", [("sym_int4", 0.3), ("asym_int4", 0.3), ("sym_int5", 0.15), ("sym_int8", 0.03),
                                         ("nf4", 0.3), ("fp4", 0.4), ("fp6", 0.15), ("fp8_e4m3", 0.08), ("fp8", 0.15),
                                         ("fp16", 0.01), ("bf16", 0.03), ("q4_k", 0.3), ("q6_k", 0.06),
                                         ("gguf_q4k_m", 0.3), ("mixed_fp4", 0.3), ("mixed_fp8", 0.05)])
def test_from_pretrained_low_bit_logits_close(tiny_llama, low_bit, tol):
    from ipex_llm.transformers import AutoModelForCausalLM
    d, ref_model = tiny_llama
    ids = torch.randint(0, 320, (2, 9))
    with torch.no_grad():
        ref = ref_model(ids).logits
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit=low_bit, optimize_model=True, use_cache=True)
    assert q.config.bigdl_transformers_low_bit == low_bit
    with torch.no_grad():
        got = q(ids).logits.float()
    assert _rel(got, ref) < tol


def test_optimized_equals_unoptimized(tiny_llama):
    """Reference test TEST/inference/test_transformers_api.py:151-169: mean relative diff < 5 %."""
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny_llama
    ids = torch.randint(0, 320, (1, 8))
    a = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True, optimize_model=False)
    b = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True, optimize_model=True)
    assert hasattr(b.model.layers[0].self_attn, "qkv_proj") and hasattr(a.model.layers[0].self_attn, "q_proj")
    with torch.no_grad():
        la, lb = a(ids).logits, b(ids).logits
    assert _rel(lb, la) < 0.05
    ga = a.generate(ids, max_new_tokens=6, do_sample=False)
    gb = b.generate(ids, max_new_tokens=6, do_sample=False)
    assert ga.shape == gb.shape == (1, 14)


@pytest.mark.parametrize("low_bit", ["sym_int4", "nf4", "sym_int5", "sym_int8", "asym_int4", "q4_k", "fp8_e5m2"])
def test_save_load_low_bit_roundtrip(tiny_llama, tmp_path, low_bit):
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny_llama
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit=low_bit)
    out = str(tmp_path / "lowbit")
    q.save_low_bit(out)
    files = set(os.listdir(out))
    assert {"config.json", "load_keys.json", "pytorch_model.bin"} <= files
    cfg = json.load(open(os.path.join(out, "config.json")))
    assert cfg["bigdl_transformers_low_bit"] == low_bit
    keys = json.load(open(os.path.join(out, "load_keys.json")))["all_checkpoint_keys"]
    sd = torch.load(os.path.join(out, "pytorch_model.bin"), weights_only=True)
    assert set(keys) == set(sd)
    w = sd["model.layers.0.mlp.down_proj.weight"]
    assert w.dim() == 1 and w.dtype == (torch.bfloat16 if low_bit == "nf4" else torch.uint8)
    q2 = AutoModelForCausalLM.load_low_bit(out)
    ids = torch.randint(0, 320, (1, 7))
    with torch.no_grad():
        assert torch.equal(q(ids).logits, q2(ids).logits)
    # error contracts (reference model.py:177-180, 597-600)
    with pytest.raises(RuntimeError, match="load_low_bit"):
        AutoModelForCausalLM.from_pretrained(out, load_in_4bit=True)
    with pytest.raises(RuntimeError, match="not a low-bit model"):
        AutoModelForCausalLM.load_low_bit(d)


def test_sharded_save_load(tiny_llama, tmp_path):
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny_llama
    q = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True)
    out = str(tmp_path / "sharded")
    q.save_low_bit(out, max_shard_size_bytes=200_000)
    assert os.path.exists(os.path.join(out, "pytorch_model.bin.index.json"))
    q2 = AutoModelForCausalLM.load_low_bit(out)
    ids = torch.randint(0, 320, (1, 5))
    with torch.no_grad():
        assert torch.equal(q(ids).logits, q2(ids).logits)


def test_modules_to_not_convert_and_bnb_config(tiny_llama):
    from ipex_llm.transformers import AutoModelForCausalLM
    from ipex_llm.transformers.low_bit_linear import LowBitLinear
    d, _ = tiny_llama
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="nf4", optimize_model=False,
                                             torch_dtype=torch.bfloat16, modules_to_not_convert=["lm_head"])
    assert isinstance(q.lm_head, torch.nn.Linear) and not isinstance(q.lm_head, LowBitLinear)
    assert isinstance(q.model.layers[0].mlp.up_proj, LowBitLinear) and q.model.layers[0].mlp.up_proj.qtype == Q.NF4
    assert q.model.norm.weight.dtype == torch.bfloat16

    class BitsAndBytesConfig:  # same attribute surface as transformers' class (bitsandbytes is not installed here)
        bnb_4bit_quant_type = "nf4"
        bnb_4bit_use_double_quant = False
        bnb_4bit_compute_dtype = torch.bfloat16
    q = AutoModelForCausalLM.from_pretrained(d, quantization_config=BitsAndBytesConfig())
    assert q.config.bigdl_transformers_low_bit == "nf4" and isinstance(q.lm_head, torch.nn.Linear)
    assert hasattr(q.model.layers[0].self_attn, "q_proj")   # optimize_model forced off for training


def test_generic_optimize_model_api(tiny_llama, tmp_path):
    """Reference flow docs/mddocs/PythonAPI/optimize.md:6-77 and TEST/convert/test_convert_model.py:91-106."""
    from ipex_llm import optimize_model
    from ipex_llm.optimize import load_low_bit, low_memory_init
    d, _ = tiny_llama
    m = transformers.AutoModelForCausalLM.from_pretrained(d)
    m = optimize_model(m, low_bit="sym_int4")
    out = str(tmp_path / "generic")
    m.save_low_bit(out)
    assert os.path.exists(os.path.join(out, "bigdl_config.json"))
    with low_memory_init():
        e = transformers.AutoModelForCausalLM.from_config(transformers.AutoConfig.from_pretrained(out))
    assert next(e.parameters()).device.type == "meta"
    e = load_low_bit(e, out)
    ids = torch.randint(0, 320, (1, 6))
    with torch.no_grad():
        assert torch.equal(m(ids).logits, e(ids).logits)
    # the documented call shape: from_pretrained on the saved directory inside the context
    with low_memory_init():
        e2 = transformers.AutoModelForCausalLM.from_pretrained(out, torch_dtype="auto")
    assert next(e2.parameters()).device.type == "meta"
    e2 = load_low_bit(e2, out)
    with torch.no_grad():
        assert torch.equal(m(ids).logits, e2(ids).logits.to(m(ids).logits.dtype))
    # and from_pretrained is back to normal afterwards
    assert next(transformers.AutoModelForCausalLM.from_pretrained(d).parameters()).device.type == "cpu"

    class Tiny(torch.nn.Module):   # a plain torch module, no HF
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(128, 64)
            self.b = torch.nn.Linear(64, 8)

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))
    t = Tiny().eval()
    x = torch.randn(3, 128)
    ref = t(x)
    t = optimize_model(t, low_bit="sym_int8")
    assert _rel(t(x), ref) < 0.05
    t.save_low_bit(str(tmp_path / "plain"))
    t2 = load_low_bit(Tiny().eval(), str(tmp_path / "plain"))
    assert torch.equal(t2(x), t(x))


def test_embedding_variants(tiny_llama):
    from ipex_llm.transformers import AutoModelForCausalLM
    from ipex_llm.transformers.embedding import CPUEmbedding, DiskEmbedding, LowBitEmbedding
    d, ref = tiny_llama
    ids = torch.randint(0, 320, (1, 6))
    q = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True, embedding_qtype="q4_k")
    assert isinstance(q.model.embed_tokens, LowBitEmbedding)
    e = q.model.embed_tokens(ids)
    assert _rel(e, ref.model.embed_tokens(ids)) < 0.15
    q = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True, cpu_embedding=True)
    assert isinstance(q.model.embed_tokens, CPUEmbedding)
    q = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True, disk_embedding=True)
    assert isinstance(q.model.embed_tokens, DiskEmbedding)
    torch.testing.assert_close(q.model.embed_tokens(ids).float(), ref.model.embed_tokens(ids), rtol=1e-2, atol=1e-3)
    with torch.no_grad():
        q(ids)


def _pack_gptq(codes_in_out, zeros_g_out, bits=4):
    in_f, out_f = codes_in_out.shape
    qweight = torch.zeros(in_f // 8, out_f, dtype=torch.int32)
    for i in range(8):
        qweight |= codes_in_out[i::8].to(torch.int32) << (4 * i)
    qzeros = torch.zeros(zeros_g_out.shape[0], out_f // 8, dtype=torch.int32)
    for i in range(8):
        qzeros |= (zeros_g_out[:, i::8].to(torch.int32) - 1) << (4 * i)
    return qweight, qzeros


def test_gptq_and_awq_repack_to_asym_int4():
    from bigdl_amd.ops import host
    from bigdl_amd.transformers.convert import convert_gptq, _AWQ_REVERSE_ORDER
    torch.manual_seed(0)
    in_f, out_f, gs = 256, 64, 128
    codes = torch.randint(0, 16, (in_f, out_f))
    zeros = torch.randint(1, 16, (in_f // gs, out_f))
    scales = (torch.rand(in_f // gs, out_f) * 0.02 + 0.01).half()
    dense = ((codes.float() - zeros.repeat_interleave(gs, 0).float()) * scales.float().repeat_interleave(gs, 0)).t()

    class G:
        pass
    g = G()
    g.bits, g.group_size, g.scales = 4, gs, scales
    g.qweight, g.qzeros = _pack_gptq(codes, zeros)
    buf, gmap = convert_gptq(g)
    assert gmap is None
    deq = host.dequantize(buf, Q.ASYM_INT4, (out_f, in_f))
    torch.testing.assert_close(deq, dense, rtol=2e-3, atol=2e-3)

    # AWQ: fields along the output dim, interleaved order, zeros stored as-is
    order = _AWQ_REVERSE_ORDER
    a = G()
    a.bits = a.w_bit = 4
    a.group_size, a.scales = gs, scales
    qw = torch.zeros(in_f, out_f // 8, dtype=torch.int32)
    qz = torch.zeros(in_f // gs, out_f // 8, dtype=torch.int32)
    for k in range(8):   # logical element k of each group of 8 sits in field order[k]
        qw |= codes[:, k::8].to(torch.int32) << (4 * order[k])
        qz |= zeros[:, k::8].to(torch.int32) << (4 * order[k])
    a.qweight, a.qzeros = qw, qz
    buf, _ = convert_gptq(a, awq=True)
    deq = host.dequantize(buf, Q.ASYM_INT4, (out_f, in_f))
    torch.testing.assert_close(deq, dense, rtol=2e-3, atol=2e-3)


def test_imatrix_reader_and_mixed_policy(tmp_path):
    from bigdl_amd.transformers.utils import get_cur_qtype_and_imatrix, load_imatrix_data
    f = tmp_path / "imatrix.dat"
    entries = {"blk.0.attn_v.weight": torch.arange(8).float(), "blk.1.ffn_down.weight": torch.ones(4)}
    with open(f, "wb") as fh:
        fh.write(struct.pack("i", len(entries)))
        for name, vals in entries.items():
            b = name.encode()
            fh.write(struct.pack("i", len(b)) + b + struct.pack("i", 2) + struct.pack("i", vals.numel()))
            fh.write(vals.numpy().tobytes())
    data = load_imatrix_data(str(f))
    assert torch.equal(data["0_v"], torch.arange(8).float() / 2) and "1_down" in data

    class Cfg:
        num_hidden_layers = 16
        model_type = "llama"
    qt, im = get_cur_qtype_and_imatrix(Q.gguf_mixed_qtype["gguf_q4k_m"], "model.layers.0.self_attn.v_proj", data, Cfg())
    assert qt == Q.Q6_K and im is not None
    qt, _ = get_cur_qtype_and_imatrix(Q.gguf_mixed_qtype["gguf_q4k_m"], "model.layers.12.self_attn.v_proj", None, Cfg())
    assert qt == Q.Q4_K
    qt, _ = get_cur_qtype_and_imatrix(Q.gguf_mixed_qtype["gguf_q4k_s"], "model.layers.1.mlp.down_proj", None, Cfg())
    assert qt == Q.Q5_K
    qt, _ = get_cur_qtype_and_imatrix(Q.IQ2_XXS, "lm_head", None, Cfg())
    assert qt == Q.SYM_INT8
    qt, _ = get_cur_qtype_and_imatrix(Q.IQ2_XXS, "model.layers.3.self_attn.v_proj", data, Cfg())
    assert qt == Q.Q2_K


def test_other_families_convert_and_run():
    """Mistral / Qwen2 / Gemma go through the same generic rewrites (merged qkv, fused MLP, RMSNorm)."""
    from ipex_llm import optimize_model
    from transformers import GemmaConfig, GemmaForCausalLM, MistralConfig, MistralForCausalLM, Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(1)
    common = dict(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                  num_key_value_heads=2, max_position_embeddings=128)
    for cfg_cls, m_cls, extra in ((MistralConfig, MistralForCausalLM, {}), (Qwen2Config, Qwen2ForCausalLM, {}),
                                  (GemmaConfig, GemmaForCausalLM, {"head_dim": 32})):
        m = m_cls(cfg_cls(**common, **extra)).eval()
        ids = torch.randint(0, 256, (1, 7))
        with torch.no_grad():
            ref = m(ids).logits
        q = optimize_model(m, low_bit="sym_int8")
        assert hasattr(q.model.layers[0].self_attn, "qkv_proj")
        with torch.no_grad():
            got = q(ids).logits
        assert _rel(got, ref) < 0.05, m_cls.__name__
        out = q.generate(ids, max_new_tokens=4, do_sample=False)
        assert out.shape == (1, 11)


def test_npu_model_import_path_takes_the_npu_keywords(tiny_llama, tmp_path):
    """A script written against ipex_llm.transformers.npu_model runs unchanged: NPU keywords are accepted, *_rtn
    qtypes land on the block formats, save_directory receives a low-bit checkpoint that load_low_bit reads back."""
    from ipex_llm.transformers.npu_model import AutoModelForCausalLM
    from ipex_llm.transformers.low_bit_linear import LowBitLinear
    d, ref_model = tiny_llama
    ids = torch.randint(0, 320, (1, 9))
    out = str(tmp_path / "npu_out")
    m = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int4_rtn", optimize_model=True, pipeline=True,
                                             max_context_len=512, max_prompt_len=128, quantization_group_size=64,
                                             transpose_value_cache=True, mixed_precision=False, device="cpu",
                                             save_directory=out, torch_dtype=torch.float16, attn_implementation="eager")
    assert m.max_context_len == 511 and m.max_prompt_len == 128
    assert isinstance(m.model.layers[0].mlp.up_proj, LowBitLinear) and m.config.bigdl_transformers_low_bit == "sym_int4"
    with torch.no_grad():
        want = m(ids).logits.float()
        assert _rel(want, ref_model(ids).logits) < 0.3
    back = AutoModelForCausalLM.load_low_bit(out, max_context_len=512, max_prompt_len=128, device="cpu")
    with torch.no_grad():
        torch.testing.assert_close(back(ids).logits.float(), want, rtol=1e-3, atol=1e-3)
    conv = AutoModelForCausalLM.optimize_npu_model(model=LlamaForCausalLM.from_pretrained(d), qtype="sym_int8_rtn",
                                                   max_context_len=256, device="cpu")
    assert conv.model.layers[0].mlp.up_proj.qtype == Q.SYM_INT8


def test_mlp_padding_and_norm_head_folding(tmp_path):
    """An intermediate size that is not a multiple of 128 (Qwen1.5-14B: 13696 is; a 200-channel test model is not) is
    zero-padded before conversion, exactly, and survives save_low_bit / load_low_bit; a NormHead output layer is folded
    into a plain linear."""
    import torch
    from torch import nn
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.models.hf_patch import fold_norm_head
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.transformers import AutoModelForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=128, hidden_size=128, intermediate_size=200, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128)
    ref = LlamaForCausalLM(cfg).eval()
    ids = torch.randint(0, 128, (1, 9))
    want = ref(ids).logits
    import copy
    model = optimize_model(copy.deepcopy(ref), low_bit="sym_int8")
    mlp = model.model.layers[0].mlp
    assert mlp.gate_proj.out_len == 256 and mlp.down_proj.in_len == 256 and type(mlp.down_proj).__name__ == "LowBitLinear"
    got = model(ids).logits
    assert torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0) > 0.999
    # the configuration describes what the modules hold (the native engine and a reloaded skeleton size their MLP from it)
    assert model.config.intermediate_size == 256 and model.config.bigdl_unpadded_intermediate_size == 200
    model.save_low_bit(str(tmp_path))
    again = AutoModelForCausalLM.load_low_bit(str(tmp_path))
    assert again.config.intermediate_size == 256
    torch.testing.assert_close(again(ids).logits, got)

    class NormHead(nn.Module):
        def __init__(self, hidden, vocab):
            super().__init__()
            self.weight = nn.Parameter(torch.randn(vocab, hidden))

        def forward(self, x):
            return nn.functional.linear(x, nn.functional.normalize(self.weight))
    holder = nn.Module()
    holder.lm_head = NormHead(16, 40)
    x = torch.randn(3, 16)
    want = holder.lm_head(x)
    fold_norm_head(holder)
    assert type(holder.lm_head) is nn.Linear
    torch.testing.assert_close(holder.lm_head(x), want)


def test_tied_embedding_model_keeps_its_low_bit_head(tmp_path):
    """tie_word_embeddings=True (Gemma, Qwen2-0.5B, Llama-3.2-1B ...): the converted head owns packed blocks and must not
    be re-tied to the dense embedding, neither after from_pretrained nor after load_low_bit / optimize.load_low_bit."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.optimize import load_low_bit, low_memory_init, optimize_model
    from bigdl_amd.transformers import AutoModelForCausalLM
    from bigdl_amd.transformers.low_bit_linear import FP4Params
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, tie_word_embeddings=True)
    m = LlamaForCausalLM(cfg).eval()
    m.save_pretrained(tmp_path / "hf")
    ids = torch.randint(0, 320, (1, 8))
    with torch.no_grad():
        ref = m(ids).logits
        q = AutoModelForCausalLM.from_pretrained(str(tmp_path / "hf"), load_in_low_bit="sym_int8")
        assert isinstance(q.lm_head.weight, FP4Params)
        out = q(ids).logits
        assert (out - ref).abs().max() / ref.abs().max() < 0.05
        q.save_low_bit(str(tmp_path / "lb"))
        r = AutoModelForCausalLM.load_low_bit(str(tmp_path / "lb"))
        assert isinstance(r.lm_head.weight, FP4Params) and torch.equal(r(ids).logits, out)
        g = optimize_model(LlamaForCausalLM.from_pretrained(str(tmp_path / "hf")).eval(), low_bit="sym_int8")
        g.save_low_bit(str(tmp_path / "lb2"))
        with low_memory_init():
            shell = LlamaForCausalLM(cfg)
        g2 = load_low_bit(shell, str(tmp_path / "lb2"))
        assert isinstance(g2.lm_head.weight, FP4Params)
        assert (g2(ids).logits - ref).abs().max() / ref.abs().max() < 0.05


def test_mixed_format_checkpoint_remembers_the_format_of_every_tensor(tmp_path):
    """mixed_fp4: each tensor settles on sym_int4 or fp4 (same packed size).  The module reports the settled format (the
    GPU kernels are dispatched on it), and save_low_bit / load_low_bit carry the per-tensor outcome."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd import qtypes as Q
    from bigdl_amd.transformers import AutoModelForCausalLM
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2)
    m = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "up_proj" in n or "o_proj" in n:
                p.copy_(torch.randn_like(p) ** 3 * 0.02)            # heavy tails: the fp format reconstructs these better
    m.save_pretrained(tmp_path / "hf")
    q = AutoModelForCausalLM.from_pretrained(str(tmp_path / "hf"), load_in_low_bit="mixed_fp4")
    kinds = {n: mod.qtype for n, mod in q.named_modules() if isinstance(mod, LowBitLinear)}
    assert set(kinds.values()) == {Q.SYM_INT4, Q.FP4}
    assert all(mod.qtype == mod.weight.qtype for mod in q.modules() if isinstance(mod, LowBitLinear))
    ids = torch.randint(0, 320, (1, 8))
    with torch.no_grad():
        out = q(ids).logits
    q.save_low_bit(str(tmp_path / "lb"))
    r = AutoModelForCausalLM.load_low_bit(str(tmp_path / "lb"))
    assert {n: mod.qtype for n, mod in r.named_modules() if isinstance(mod, LowBitLinear)} == kinds
    with torch.no_grad():
        assert torch.equal(r(ids).logits, out)


@pytest.mark.parametrize("tied", [False, True])
@pytest.mark.parametrize("kw", [dict(optimize_model=False), dict(modules_to_not_convert=["lm_head"]), dict(mixed_precision=True),
                                dict(disk_embedding=True), dict(cpu_embedding=True)],
                         ids=["no_opt", "dense_head", "mixed_precision", "disk_embedding", "cpu_embedding"])
def test_low_bit_checkpoint_describes_how_it_was_converted(tmp_path, tied, kw):
    """A low-bit checkpoint reloads into the module tree it was saved from without the caller repeating
    ``optimize_model`` / ``modules_to_not_convert`` / ``mixed_precision`` (they are kept in its config; keyword arguments of
    the same names still override), and an embedding that lives in a file is written out whole."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.transformers import AutoModelForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, tie_word_embeddings=tied)
    LlamaForCausalLM(cfg).eval().save_pretrained(tmp_path / "hf")
    ids = torch.randint(0, 320, (1, 8))
    q = AutoModelForCausalLM.from_pretrained(str(tmp_path / "hf"), load_in_low_bit="sym_int4", **kw)
    with torch.no_grad():
        out = q(ids).logits
    q.save_low_bit(str(tmp_path / "lb"))
    r = AutoModelForCausalLM.load_low_bit(str(tmp_path / "lb"), **({"cpu_embedding": True} if kw.get("cpu_embedding") else {}))
    with torch.no_grad():
        assert torch.equal(r(ids).logits, out)
    assert [n for n, _ in r.named_modules()] == [n for n, _ in q.named_modules()]


def test_load_override_is_recorded_for_the_next_save(tmp_path):
    """A checkpoint from before the conversion options were recorded needs ``optimize_model=False`` spelled out at load;
    saving the loaded model again must record that, so the copy reloads with no arguments."""
    import json
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.transformers import AutoModelForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2)
    LlamaForCausalLM(cfg).eval().save_pretrained(tmp_path / "hf")
    q = AutoModelForCausalLM.from_pretrained(str(tmp_path / "hf"), load_in_low_bit="sym_int4", optimize_model=False)
    ids = torch.randint(0, 320, (1, 8))
    with torch.no_grad():
        out = q(ids).logits
    q.save_low_bit(str(tmp_path / "old"))
    cfg_file = tmp_path / "old" / "config.json"
    doc = json.loads(cfg_file.read_text())
    del doc["bigdl_convert_options"]                               # what an older writer produced
    cfg_file.write_text(json.dumps(doc))
    with pytest.raises(Exception):
        AutoModelForCausalLM.load_low_bit(str(tmp_path / "old"))  # default optimize_model=True builds merged qkv
    r = AutoModelForCausalLM.load_low_bit(str(tmp_path / "old"), optimize_model=False)
    assert r.config.bigdl_convert_options["optimize_model"] is False
    r.save_low_bit(str(tmp_path / "new"))
    again = AutoModelForCausalLM.load_low_bit(str(tmp_path / "new"))
    with torch.no_grad():
        assert torch.equal(again(ids).logits, out)


def test_xpu_device_alias_runs_reference_style_scripts():
    """``enable_xpu_alias``: "xpu" in ``.to(...)``, factory functions, ``torch.device`` objects and ``torch.xpu.*`` calls
    resolves to the device of this build (here the CPU), and goes away again when disabled."""
    import ipex_llm
    target = ipex_llm.enable_xpu_alias("cpu")
    try:
        assert target == "cpu" and ipex_llm.enable_xpu_alias() == "cpu"          # second call: already on
        x = torch.zeros(2, 3, device="xpu")
        y = torch.tensor([1, 2], device=torch.device("xpu", 0)).to("xpu:0")
        m = torch.nn.Linear(3, 2).to("xpu").to(device="xpu", dtype=torch.float32)
        out = m(x.to("xpu"))
        assert x.device.type == y.device.type == out.device.type == "cpu" and m.xpu() is m
        assert torch.randn(2).xpu().device.type == "cpu" and torch.arange(3, device="xpu").sum().item() == 3
        torch.xpu.synchronize()
        torch.xpu.empty_cache()
        assert torch.xpu.is_available() and torch.xpu.max_memory_allocated() == 0 and torch.xpu.device_count() == 1
    finally:
        ipex_llm.disable_xpu_alias()
    with pytest.raises(Exception):
        torch.zeros(1, device="xpu")
