"""HF-facing API on the GPU: module path (LowBitLinear + bound forwards) vs native engine vs fp32 reference."""
import pytest
import torch
import transformers
from transformers import LlamaConfig, LlamaForCausalLM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, head_dim=64, max_position_embeddings=512,
                      tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    d = str(tmp_path_factory.mktemp("tiny_llama_gpu"))
    m.save_pretrained(d)
    return d, m


def _rel(a, b):
    return ((a - b).abs().mean() / b.abs().mean()).item()


@pytest.mark.parametrize("low_bit", ["sym_int4", "asym_int4", "fp8_e5m2", "nf4", "sym_int8", "q4_k", "q5_k", "q6_k", "fp16"])
def test_gpu_logits_match_cpu_path(tiny, low_bit):
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit=low_bit, optimize_model=True)
    ids = torch.randint(0, 320, (2, 9))
    with torch.no_grad():
        cpu = q(ids).logits.float()
    q = q.half().to("cuda")
    if low_bit in ("q4_k", "q5_k", "sym_int4"):
        assert q.model.layers[0].mlp.down_proj.weight.qlayout == "t16"       # fused kernels, not dequantize + GEMM
    with torch.no_grad():
        gpu = q(ids.cuda()).logits.float().cpu()
    # reference test TEST/inference_gpu/test_transformers_api_final_logits.py:59-68 uses a mismatch fraction;
    # fp16 activations on the device vs fp32 on the host: compare in relative mean error
    assert _rel(gpu, cpu) < 0.03, low_bit
    # prefill-sized input takes the dequant + dense GEMM path
    ids = torch.randint(0, 320, (1, 100))
    with torch.no_grad():
        a = q(ids.cuda()).logits.float().cpu()
    q.to("cpu")
    q.float()
    with torch.no_grad():
        b = q(ids).logits.float()
    assert _rel(a, b) < 0.03


def test_native_generate_matches_module_path(tiny):
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True).half().to("cuda")
    ids = torch.randint(0, 320, (1, 20)).cuda()
    out_native = q.generate(ids, max_new_tokens=10, do_sample=False)
    assert hasattr(q, "_native_engine") and q.first_cost > 0 and q.rest_cost_mean > 0
    out_hf = q._hf_generate(ids, max_new_tokens=10, do_sample=False)
    assert out_native.shape == out_hf.shape == (1, 30)
    # same weights, same fp16 math up to kernel order: the greedy continuations agree on (nearly) every token
    assert (out_native == out_hf).float().mean() > 0.9
    # engine logits vs module-path logits for the prompt
    eng = q._native_engine
    eng.cache.reset()
    lg = eng.prefill(ids, last_only=False)
    with torch.no_grad():
        ref = q(ids[:, :9]).logits.float()
    assert _rel(lg[:, :9].float(), ref) < 0.02


def test_padded_mlp_runs_on_the_engine_and_round_trips(tmp_path):
    """An intermediate size that is not a multiple of 128 (320 here) is zero-padded at conversion; the engine's MLP buffers
    follow the padded width, generation matches the module path, and a checkpoint saved from the GPU reloads."""
    from ipex_llm.transformers import AutoModelForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=320, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=64, max_position_embeddings=512, tie_word_embeddings=False)
    LlamaForCausalLM(cfg).eval().save_pretrained(str(tmp_path / "src"))
    q = AutoModelForCausalLM.from_pretrained(str(tmp_path / "src"), load_in_4bit=True).half().to("cuda")
    assert q.config.intermediate_size == 384
    ids = torch.randint(0, 320, (1, 20)).cuda()
    out_native = q.generate(ids, max_new_tokens=10, do_sample=False)
    assert hasattr(q, "_native_engine") and q._native_engine.cfg.intermediate_size == 384
    out_hf = q._hf_generate(ids, max_new_tokens=10, do_sample=False)
    assert (out_native == out_hf).float().mean() > 0.9
    q.save_low_bit(str(tmp_path / "low"))
    again = AutoModelForCausalLM.load_low_bit(str(tmp_path / "low")).half().to("cuda")
    assert torch.equal(again.generate(ids, max_new_tokens=10, do_sample=False), out_native)


@pytest.mark.parametrize("side", ["left", "right"])
def test_padded_batch_generates_on_the_engine(tiny, side):
    """A batch with padding goes to the engine (one cache slot per row, prompts packed without their pad tokens):
    every row continues exactly as it does alone, and as HuggingFace's generate continues the left-padded batch."""
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_4bit=True).half().to("cuda")
    q.generation_config.pad_token_id = 0
    torch.manual_seed(1)
    lens = [20, 7, 13, 20]
    S = max(lens)
    ids = torch.zeros(4, S, dtype=torch.long)
    mask = torch.zeros(4, S, dtype=torch.long)
    rows = [torch.randint(3, 320, (n,)) for n in lens]
    for i, r in enumerate(rows):
        sl = slice(S - len(r), S) if side == "left" else slice(0, len(r))
        ids[i, sl], mask[i, sl] = r, 1
    ids, mask = ids.cuda(), mask.cuda()
    out = q.generate(ids, attention_mask=mask, max_new_tokens=8, do_sample=False)
    eng = q._native_engine
    assert eng.cache.slot_len is not None and sorted(eng.cache.slot_len) == sorted(n + 7 for n in lens)   # engine, slot mode
    assert out.shape == (4, S + 8) and torch.equal(out[:, :S], ids)
    for i, r in enumerate(rows):
        alone = q.generate(r.view(1, -1).cuda(), max_new_tokens=8, do_sample=False)
        assert (alone[0, len(r):] == out[i, S:]).float().mean() >= 0.85, (i, alone[0, len(r):], out[i, S:])
    if side == "left":
        hf = q._hf_generate(ids, attention_mask=mask, max_new_tokens=8, do_sample=False)
        assert (hf[:, S:] == out[:, S:]).float().mean() > 0.85


@pytest.mark.parametrize("rows,bias", [(1, False), (5, True), (16, True), (17, True)])
def test_fp16_linear_decode_rows_use_the_streaming_kernel(rows, bias):
    """FP16Linear with <= 16 rows runs hip.gemv_f16 (bias in the epilogue); more rows take the dense product."""
    from bigdl_amd.ops import hip
    from bigdl_amd.transformers.low_bit_linear import FP16Linear
    torch.manual_seed(rows)
    lin = FP16Linear(512, 384, bias=bias)
    lin.weight.data = (torch.randn(384, 512) * 0.05).half()
    if bias:
        lin.bias.data = (torch.randn(384) * 0.1).half()
    lin = lin.cuda()
    x = torch.randn(1, rows, 512).half().cuda()
    seen = []
    real = hip.gemv_f16
    hip.gemv_f16 = lambda *a, **k: (seen.append(1), real(*a, **k))[1]
    try:
        y = lin(x)
    finally:
        hip.gemv_f16 = real
    assert len(seen) == (1 if rows <= 16 else 0)
    ref = x.float() @ lin.weight.float().t() + (lin.bias.float() if bias else 0)
    assert y.shape == (1, rows, 384) and y.dtype == torch.float16
    torch.testing.assert_close(y.float(), ref, rtol=5e-3, atol=5e-3)


def test_save_from_gpu_and_reload(tiny, tmp_path):
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="asym_int4")
    cpu_sd = {k: v.clone() for k, v in q.state_dict().items()}
    q = q.to("cuda")
    out = str(tmp_path / "from_gpu")
    q.save_low_bit(out)            # device tiles -> host blocks on the way out
    assert next(q.parameters()).device.type == "cuda"
    sd = torch.load(out + "/pytorch_model.bin", weights_only=True)
    for k, v in sd.items():
        assert torch.equal(v, cpu_sd[k]), k
    q2 = AutoModelForCausalLM.load_low_bit(out).to("cuda")
    ids = torch.randint(0, 320, (1, 6)).cuda()
    with torch.no_grad():
        assert torch.equal(q(ids).logits, q2(ids).logits)


def test_layer_level_equivalence(tiny):
    """Reference TEST/inference_gpu/test_transformers_api_attention.py / _mlp.py / _RMSNorm.py: outputs of the
    optimized attention / MLP / RMSNorm equal the unoptimized modules' for identical inputs."""
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny
    a = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8", optimize_model=False).half().to("cuda")
    b = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8", optimize_model=True).half().to("cuda")
    x = (torch.randn(1, 5, 256) * 0.5).half().cuda()
    la, lb = a.model.layers[1], b.model.layers[1]
    with torch.no_grad():
        torch.testing.assert_close(lb.input_layernorm(x), la.input_layernorm(x), rtol=2e-2, atol=2e-3)
        torch.testing.assert_close(lb.mlp(x), la.mlp(x), rtol=2e-2, atol=5e-3)
        pos = torch.arange(5, device="cuda")[None]
        pe = a.model.rotary_emb(x, pos)
        oa = la.self_attn(x, position_embeddings=pe, attention_mask=None)[0]
        ob = lb.self_attn(x.clone(), position_embeddings=pe, attention_mask=None)[0]
        torch.testing.assert_close(ob, oa, rtol=2e-2, atol=5e-3)


def test_lookup_and_speculative_on_native_engine(tiny):
    """Multi-token verification steps of the native engine (graph replay of q_len tokens + KV counter rollback)
    reproduce its own single-token greedy decoding."""
    from ipex_llm.transformers import AutoModelForCausalLM
    from bigdl_amd.transformers.speculative import attach_speculative
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8").half().to("cuda")
    torch.manual_seed(11)
    ids = torch.randint(0, 320, (1, 16)).repeat(1, 3).cuda()
    ref = q.generate(ids, max_new_tokens=24, do_sample=False)
    out = q.generate(ids, max_new_tokens=24, lookahead=3, max_matching_ngram_size=2)
    assert out.shape == ref.shape and q.n_drafted > 0
    assert (out == ref).float().mean() > 0.9
    draft = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int4").half().to("cuda")
    attach_speculative(q, draft)
    out2 = q.generate(ids, max_new_tokens=24, max_step_draft=4, th_stop_draft=0.0, min_step_draft=1)
    assert out2.shape == ref.shape and q.n_drafted > 0 and len(q.verify_time) == len(q.draft_time)
    assert (out2 == ref).float().mean() > 0.9
    out3 = q.generate(ids, max_new_tokens=12, do_sample=True, top_k=10)
    assert out3.shape[1] == ids.shape[1] + 12
    # a draft that runs one of its two layers (layer skipping on the shared cache): still the target's own tokens
    full_pair = q._spec_engine
    out4 = q.generate(ids, max_new_tokens=24, max_step_draft=4, th_stop_draft=0.0, min_step_draft=1, draft_layers=1)
    assert q._spec_engine is not full_pair and len(q._spec_engine.draft.layers) == 1 and q._spec_engine.draft.cache_layer == [0]
    assert out4.shape == ref.shape and (out4 == ref).float().mean() > 0.9 and q.n_drafted > 0


def test_fp8_kv_cache_module_path(tiny, monkeypatch):
    from ipex_llm.transformers import AutoModelForCausalLM
    from bigdl_amd.transformers import kv
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8").half().to("cuda")
    ids = torch.randint(0, 320, (1, 40)).cuda()
    with torch.no_grad():
        ref = q(ids, use_cache=True)
        monkeypatch.setenv("IPEX_LLM_QUANTIZE_KV_CACHE", "1")
        out = q(ids, use_cache=True)
        assert isinstance(out.past_key_values, kv.DynamicFp8Cache)
        nxt = ref.logits[:, -1:].argmax(-1)
        a = q(nxt, past_key_values=ref.past_key_values, use_cache=True).logits.float()
        b = q(nxt, past_key_values=out.past_key_values, use_cache=True).logits.float()
    assert _rel(b, a) < 0.12      # e5m2 keeps 2 mantissa bits


def test_fp8_kv_cache_read_as_bytes_on_the_module_path(tiny, monkeypatch):
    """fp16 models whose attention goes through the registry entry get the fp8 cache that is never expanded: the flash
    kernel reads K / V^T bytes in place (prompt, a 3-token step, single-token steps); logits equal those of the
    expanding fp8 cache (same e5m2 values either way), generation runs on it."""
    from ipex_llm.transformers import AutoModelForCausalLM
    from bigdl_amd.transformers import kv
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8").half().to("cuda")
    monkeypatch.setenv("IPEX_LLM_QUANTIZE_KV_CACHE", "1")
    ids = torch.randint(0, 320, (2, 40)).cuda()
    more = torch.randint(0, 320, (2, 3)).cuda()
    res = {}
    for raw in ("1", "0"):
        monkeypatch.setenv("BIGDL_FP8_RAW_CACHE", raw)
        with torch.no_grad():
            out = q(ids, use_cache=True)
            pkv = out.past_key_values
            assert isinstance(pkv, kv.DynamicFp8RawCache) == (raw == "1") and isinstance(pkv, kv.DynamicFp8Cache)
            step3 = q(more, past_key_values=pkv, use_cache=True).logits.float()
            nxt = step3[:, -1:].argmax(-1)
            step1 = q(nxt, past_key_values=pkv, use_cache=True).logits.float()
        assert pkv.get_seq_length() == 44
        res[raw] = (out.logits.float(), step3, step1)
    assert pkv.layers[0].keys.dtype == torch.float16               # (last pass: the expanding cache hands back fp16 copies)
    for a, b in zip(res["1"], res["0"]):
        assert _rel(a, b) < 2e-2
    monkeypatch.setenv("BIGDL_FP8_RAW_CACHE", "1")
    gen = q._hf_generate(ids, max_new_tokens=6, do_sample=False)
    assert gen.shape == (2, 46)


@pytest.mark.parametrize("family", ["mixtral", "qwen2_moe"])
def test_moe_models_module_path_and_native_engine(family, tmp_path):
    """Fused HF experts -> LowBitExperts; routed GEMV decode (module path and hipGraph engine) vs the CPU model."""
    import copy
    from transformers import MixtralConfig, MixtralForCausalLM, Qwen2MoeConfig, Qwen2MoeForCausalLM
    from bigdl_amd.optimize import optimize_model
    torch.manual_seed(0)
    if family == "mixtral":
        cfg = MixtralConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                            num_attention_heads=4, num_key_value_heads=2, num_local_experts=4, num_experts_per_tok=2,
                            head_dim=64, max_position_embeddings=512)
        m = MixtralForCausalLM(cfg).eval()
    else:
        cfg = Qwen2MoeConfig(vocab_size=320, hidden_size=256, intermediate_size=512, moe_intermediate_size=128,
                             shared_expert_intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                             num_key_value_heads=2, num_experts=8, num_experts_per_tok=2, max_position_embeddings=512)
        m = Qwen2MoeForCausalLM(cfg).eval()
    q = optimize_model(copy.deepcopy(m), low_bit="sym_int8")
    ids = torch.randint(0, 320, (1, 9))
    with torch.no_grad():
        cpu = q(ids).logits.float()
    q = q.half().to("cuda")
    with torch.no_grad():
        gpu = q(ids.cuda()).logits.float().cpu()          # 9 tokens x 2 slots: routed GEMV path
        long_ids = torch.randint(0, 320, (1, 70))
        a = q(long_ids.cuda()).logits.float().cpu()       # bucketed dense path
    assert _rel(gpu, cpu) < 0.03
    out_native = q.generate(ids.cuda(), max_new_tokens=8, do_sample=False)
    assert hasattr(q, "_native_engine")
    out_hf = q._hf_generate(ids.cuda(), max_new_tokens=8, do_sample=False)
    assert out_native.shape == out_hf.shape
    assert (out_native == out_hf).float().mean() > 0.85
    eng = q._native_engine
    eng.cache.reset()
    lg = eng.prefill(long_ids.cuda(), last_only=False)
    # the module path's lm_head keeps only the last position for long inputs
    assert a.shape[1] == 1 and _rel(lg[:, -1:].float().cpu(), a) < 0.03


def test_rwkv_model_uses_recurrence_kernels():
    """HF RWKV-4: low-bit linears + the wkv / time-shift kernels reproduce the fp32 CPU model, with and without state."""
    import copy
    from transformers import RwkvConfig, RwkvForCausalLM
    from bigdl_amd.optimize import optimize_model
    torch.manual_seed(0)
    cfg = RwkvConfig(vocab_size=320, hidden_size=256, num_hidden_layers=2, attention_hidden_size=256, intermediate_size=512,
                     context_length=128)
    m = RwkvForCausalLM(cfg).eval()
    ids = torch.randint(0, 320, (1, 12))
    q = optimize_model(copy.deepcopy(m), low_bit="sym_int8")
    assert hasattr(q.rwkv.blocks[0].attention, "_hf_forward")
    with torch.no_grad():
        ref = q(ids).logits.float()                      # CPU path of the same quantized model
    q = q.to("cuda")
    with torch.no_grad():
        out = q(ids.cuda(), use_cache=True)
        assert _rel(out.logits.float().cpu(), ref) < 0.02
        nxt = out.logits[:, -1:].argmax(-1)
        step = q(nxt, state=out.state, use_cache=True).logits.float().cpu()
        full = q(torch.cat([ids.cuda(), nxt], 1)).logits[:, -1:].float().cpu()
    assert _rel(step, full) < 0.02


def test_left_padded_batch_on_module_path(tiny):
    """Padded batches go through the HF forward with an additive mask -> masked flash-attention kernel."""
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8").half().to("cuda")
    q.generation_config.pad_token_id = 0
    torch.manual_seed(3)
    a, b = torch.randint(1, 320, (1, 12)).cuda(), torch.randint(1, 320, (1, 7)).cuda()
    ids = torch.zeros(2, 12, dtype=torch.long, device="cuda")
    ids[0] = a[0]
    ids[1, 5:] = b[0]
    mask = torch.ones(2, 12, dtype=torch.long, device="cuda")
    mask[1, :5] = 0
    with torch.no_grad():
        batch = q(input_ids=ids, attention_mask=mask).logits.float()
        one_a = q(input_ids=a).logits.float()
        one_b = q(input_ids=b).logits.float()
    assert _rel(batch[0, -1], one_a[0, -1]) < 0.02 and _rel(batch[1, -1], one_b[0, -1]) < 0.02
    out = q.generate(ids, attention_mask=mask, max_new_tokens=6, do_sample=False)
    solo = q.generate(b, max_new_tokens=6, do_sample=False)
    assert (out[1, 12:] == solo[0, 7:]).float().mean() > 0.8


@pytest.mark.parametrize("family", ["gpt2", "gpt_neox", "phi", "starcoder2", "opt", "qwen3", "whisper"])
def test_attention_registry_reaches_flash_kernel(family, monkeypatch):
    """Families that dispatch through HF's attention registry (no per-family forward here) run the MFMA flash kernel
    on the GPU and reproduce the CPU logits of the same quantized model."""
    import copy
    from transformers import AutoConfig, AutoModelForCausalLM, WhisperConfig, WhisperForConditionalGeneration
    from bigdl_amd.ops import hip
    from bigdl_amd.optimize import optimize_model
    torch.manual_seed(0)
    common = dict(vocab_size=320, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=256)
    extra = {"gpt2": dict(n_embd=256, n_layer=2, n_head=4, n_positions=256), "gpt_neox": dict(intermediate_size=512),
             "phi": dict(intermediate_size=512), "starcoder2": dict(intermediate_size=512, num_key_value_heads=2),
             "opt": dict(ffn_dim=512, word_embed_proj_dim=256), "gptj": dict(n_embd=256, n_layer=2, n_head=4, n_positions=256, rotary_dim=32),
             "qwen3": dict(intermediate_size=512, num_key_value_heads=2, head_dim=64)}
    calls = {"n": 0}
    real = hip.attn_prefill

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(hip, "attn_prefill", counted)
    if family == "whisper":
        cfg = WhisperConfig(vocab_size=320, d_model=256, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                            decoder_attention_heads=4, encoder_ffn_dim=512, decoder_ffn_dim=512, num_mel_bins=16,
                            max_source_positions=32, max_target_positions=32, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                            decoder_start_token_id=1)
        q = optimize_model(WhisperForConditionalGeneration(cfg).eval(), low_bit="sym_int8")
        feats, dec = torch.randn(1, 16, 64), torch.randint(3, 320, (1, 7))
        with torch.no_grad():
            ref = q(input_features=feats, decoder_input_ids=dec).logits.float()
            q = q.half().to("cuda")
            out = q(input_features=feats.half().cuda(), decoder_input_ids=dec.cuda()).logits.float().cpu()
    else:
        cfg = AutoConfig.for_model(family, **{**common, **extra[family]})
        q = optimize_model(AutoModelForCausalLM.from_config(cfg).eval().float(), low_bit="sym_int8")
        ids = torch.randint(0, 320, (2, 9))
        with torch.no_grad():
            ref = q(ids).logits.float()
            q = q.half().to("cuda")
            out = q(ids.cuda()).logits.float().cpu()
            gen = q.generate(ids[:1].cuda(), max_new_tokens=4, do_sample=False)
        assert gen.shape == (1, 13)
    assert calls["n"] >= 2, "the flash kernel was not reached"
    assert _rel(out, ref) < 0.03


def test_granite_multipliers_and_streamed_prompt(tmp_path):
    """Granite checkpoints scale the embedding, the attention scores and the logits: the native engine takes the first
    three from the config (engine logits = HF module logits); a residual multiplier sends the model to the HF path.
    A streamer with skip_prompt=True sees exactly the generated tokens (the prompt is put first, as HF generate does)."""
    from transformers import GraniteConfig, GraniteForCausalLM, TextIteratorStreamer
    from ipex_llm.transformers import AutoModelForCausalLM
    from bigdl_amd.models.hf_patch import native_engine_supported
    torch.manual_seed(0)
    cfg = GraniteConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, max_position_embeddings=512, embedding_multiplier=6.0,
                        attention_multiplier=0.07, logits_scaling=4.0, residual_multiplier=1.0, tie_word_embeddings=False)
    d = str(tmp_path / "granite")
    GraniteForCausalLM(cfg).eval().save_pretrained(d)
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8").half().to("cuda")
    assert native_engine_supported(q)
    ids = torch.randint(0, 320, (1, 12)).cuda()
    out = q.generate(ids, max_new_tokens=6, do_sample=False)
    eng = q._native_engine
    eng.cache.reset()
    lg = eng.prefill(ids, last_only=False)[:, :9].float()
    with torch.no_grad():
        ref = q(ids[:, :9]).logits.float()
    assert _rel(lg, ref) < 0.02

    class Tok:          # the streamer only needs decode()
        def decode(self, toks, **kw):
            return "".join(f"<{int(t)}>" for t in toks)
    st = TextIteratorStreamer(Tok(), skip_prompt=True)
    out2 = q.generate(ids, max_new_tokens=6, do_sample=False, streamer=st)
    text = "".join(st)
    assert text == Tok().decode(out2[0, 12:])
    assert torch.equal(out, out2)

    cfg.residual_multiplier = 0.5
    d2 = str(tmp_path / "granite_res")
    GraniteForCausalLM(cfg).eval().save_pretrained(d2)
    q2 = AutoModelForCausalLM.from_pretrained(d2, load_in_low_bit="sym_int8").half().to("cuda")
    assert not native_engine_supported(q2)
    assert q2.generate(ids, max_new_tokens=3, do_sample=False).shape == (1, 15)


def test_qwen2_vl_prompt_with_image_on_the_engine():
    """Image features + 3-axis rotary positions reach the engine: prompt logits and the following decode steps agree
    with the HuggingFace forward of the same (low-bit) model."""
    import copy
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    from bigdl_amd.models.hf_patch import get_native_engine, language_model_inputs, native_engine_supported
    from bigdl_amd.optimize import optimize_model
    cfg = Qwen2VLConfig(
        text_config=dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=2, max_position_embeddings=2048, bos_token_id=1, eos_token_id=2,
                         rope_parameters=dict(rope_type="default", mrope_section=[8, 12, 12], rope_theta=10000.0)),
        vision_config=dict(depth=2, embed_dim=64, hidden_size=256, num_heads=4, mlp_ratio=2, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3),
        image_token_id=500, video_token_id=501, vision_start_token_id=502, vision_end_token_id=503)
    torch.manual_seed(0)
    model = Qwen2VLForConditionalGeneration(cfg).eval()
    model = optimize_model(model, low_bit="sym_int8").half().cuda()
    assert native_engine_supported(model)
    grid = torch.tensor([[1, 8, 8]], device="cuda")
    n_img = 8 * 8 // 4
    ids = torch.tensor([[5, 6, 502] + [500] * n_img + [503, 7, 8, 9, 10, 11]], device="cuda")
    pix = torch.randn(64, 3 * 2 * 14 * 14, device="cuda", dtype=torch.float16)
    with torch.no_grad():
        ref = model(input_ids=ids, pixel_values=pix, image_grid_thw=grid, use_cache=True,
                    mm_token_type_ids=(ids == 500).long())
    embeds, pos = language_model_inputs(model, ids, dict(pixel_values=pix, image_grid_thw=grid))
    assert embeds.shape == (1, ids.shape[1], 256) and pos.shape == (3, 1, ids.shape[1])
    assert int(pos.max()) + 1 < ids.shape[1]                     # image tokens share positions: a negative shift follows
    eng = get_native_engine(model)
    assert eng.cfg.mrope_section == (8, 12, 12)
    eng.ensure_cache(1, 256)
    eng.cache.reset()
    logits = eng.prefill(ids, inputs_embeds=embeds, mrope_positions=pos)[:, -1].float()
    want = ref.logits[:, -1].float()
    assert torch.nn.functional.cosine_similarity(logits.flatten(), want.flatten(), dim=0) > 0.999
    assert eng._rope_shift == int(pos.max()) + 1 - ids.shape[1]
    # two decode steps on both sides, fed the same tokens
    past, tok = ref.past_key_values, want.argmax(-1).view(1, 1)
    for _ in range(2):
        with torch.no_grad():
            step = model(input_ids=tok, past_key_values=past, use_cache=True)
        past = step.past_key_values
        lg, _ = eng.decode(tok, 1)
        assert torch.nn.functional.cosine_similarity(lg.float().flatten(), step.logits[:, -1].float().flatten(), dim=0) > 0.999
        tok = step.logits[:, -1].argmax(-1).view(1, 1)
    # and through the front door
    out = model.generate(input_ids=ids, pixel_values=pix, image_grid_thw=grid, max_new_tokens=4, do_sample=False)
    assert out.shape == (1, ids.shape[1] + 4) and model.first_cost > 0


def test_mllama_cross_attention_runs_on_the_flash_kernel():
    """Cross-attention layers of Mllama (queries from text, keys / values from the vision tower, additive
    cross-attention mask) go through ``attn_prefill``; logits agree with the eager attention of the same weights."""
    import copy
    from transformers import MllamaConfig, MllamaForConditionalGeneration
    from bigdl_amd.ops import hip
    from bigdl_amd.optimize import optimize_model
    cfg = MllamaConfig(
        vision_config=dict(hidden_size=64, attention_heads=4, num_hidden_layers=2, num_global_layers=1, intermediate_size=128,
                           image_size=28, patch_size=14, max_num_tiles=1, vision_output_dim=128,
                           intermediate_layers_indices=[0], supported_aspect_ratios=[[1, 1]]),
        text_config=dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=4,
                         num_key_value_heads=2, cross_attention_layers=[1, 3], max_position_embeddings=512,
                         rope_parameters=dict(rope_type="default", rope_theta=10000.0), pad_token_id=0, bos_token_id=1,
                         eos_token_id=2),
        image_token_index=300)
    torch.manual_seed(0)
    plain = MllamaForConditionalGeneration(cfg).eval()
    eager = copy.deepcopy(plain).half().cuda()
    eager.config._attn_implementation = "eager"
    eager.config.text_config._attn_implementation = "eager"
    model = optimize_model(plain, low_bit="bf16").half().cuda()
    ids = torch.tensor([[1, 300, 5, 6, 7, 8, 9, 10]], device="cuda")
    kw = dict(pixel_values=torch.randn(1, 1, 1, 3, 28, 28, device="cuda", dtype=torch.float16),
              aspect_ratio_ids=torch.tensor([[1]], device="cuda"), aspect_ratio_mask=torch.ones(1, 1, 1, dtype=torch.long, device="cuda"),
              cross_attention_mask=torch.ones(1, 8, 1, 1, dtype=torch.long, device="cuda"))
    calls = []
    orig = hip.attn_prefill

    def spy(q, k, v, **k2):
        calls.append((q.shape[2], k.shape[2], k2.get("causal")))
        return orig(q, k, v, **k2)
    hip.attn_prefill = spy
    try:
        with torch.no_grad():
            got = model(input_ids=ids, **kw).logits.float()
    finally:
        hip.attn_prefill = orig
    with torch.no_grad():
        want = eager(input_ids=ids, **kw).logits.float()
    cross = [c for c in calls if c[1] != c[0]]
    assert len(cross) == 2 and all(c[2] is False for c in cross), calls      # two cross-attention layers, non-causal
    assert torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0) > 0.999


def test_siglip_tower_72_wide_heads_and_internvl_wrapper_on_gpu(monkeypatch):
    """The image encoders of MiniCPM-V (SigLIP, 72-channel heads: reference models/minicpmv.py:44-90) and InternVL
    (models/internvl.py:170) on the GPU module path: attention through the flash kernel (72-wide tiles included), logits
    equal to the CPU run of the same quantized model."""
    import copy
    from transformers import AutoModelForImageTextToText, InternVLConfig, SiglipVisionConfig, SiglipVisionModel
    from bigdl_amd.ops import hip
    from bigdl_amd.optimize import optimize_model
    calls = {"n": 0, "dims": set()}
    real = hip.attn_prefill

    def counted(q, *a, **k):
        calls["n"] += 1
        calls["dims"].add(q.shape[-1])
        return real(q, *a, **k)
    monkeypatch.setattr(hip, "attn_prefill", counted)
    torch.manual_seed(0)
    tower = SiglipVisionModel(SiglipVisionConfig(hidden_size=288, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                                 image_size=56, patch_size=14)).eval()
    q = optimize_model(tower, low_bit="sym_int8")
    x = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        ref = q(pixel_values=x).last_hidden_state.float()
        out = copy.deepcopy(q).half().to("cuda")(pixel_values=x.half().cuda()).last_hidden_state.float().cpu()
    assert 72 in calls["dims"], "the 72-wide flash kernel was not used"
    assert _rel(out, ref) < 0.03
    text = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=256)
    cfg = InternVLConfig(vision_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                            image_size=[28, 28], patch_size=[14, 14]),
                         text_config=dict(model_type="qwen2", **text), image_token_id=319, downsample_ratio=0.5)
    m = optimize_model(AutoModelForImageTextToText.from_config(cfg).eval(), low_bit="sym_int8")
    ids, px = torch.tensor([[1, 5, 319, 7, 8, 9]]), torch.randn(1, 3, 28, 28)
    n0 = calls["n"]
    with torch.no_grad():
        ref = m(input_ids=ids, pixel_values=px).logits.float()
        out = copy.deepcopy(m).half().to("cuda")(input_ids=ids.cuda(), pixel_values=px.half().cuda()).logits.float().cpu()
    assert calls["n"] > n0
    assert _rel(out, ref) < 0.03


def test_t5_seq2seq_low_bit_on_gpu(tmp_path):
    """flan-t5 style encoder-decoder, every linear low-bit (``wo`` and the head included), fp16 on the GPU."""
    from transformers import T5Config, T5ForConditionalGeneration
    from ipex_llm.transformers import AutoModelForSeq2SeqLM
    torch.manual_seed(0)
    cfg = T5Config(vocab_size=320, d_model=256, d_kv=64, d_ff=512, num_layers=2, num_decoder_layers=2, num_heads=4,
                   feed_forward_proj="gated-gelu", decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
                   tie_word_embeddings=False)
    m = T5ForConditionalGeneration(cfg).eval()
    m.save_pretrained(tmp_path)
    ids = torch.randint(3, 320, (2, 12))
    dec = torch.randint(3, 320, (2, 5))
    with torch.no_grad():
        ref = m(input_ids=ids, decoder_input_ids=dec).logits
        q = AutoModelForSeq2SeqLM.from_pretrained(str(tmp_path), load_in_low_bit="sym_int8").half().to("cuda")
        out = q(input_ids=ids.cuda(), decoder_input_ids=dec.cuda()).logits.float().cpu()
        gen = q.generate(ids.cuda(), max_new_tokens=6, do_sample=False, use_cache=True)
    assert _rel(out, ref) < 0.05
    assert gen.shape[0] == 2 and gen.shape[1] <= 7


def test_streaming_turns_on_a_bounded_cache_through_the_model_engine(tiny):
    """The streaming-llm application flow: ``model.native_engine()``, turns appended to one fp16 cache, attention-sink
    eviction whenever the next turn would not fit; the cache never outgrows its capacity and tokens keep coming."""
    from ipex_llm.transformers import AutoModelForCausalLM
    d, _ = tiny
    q = AutoModelForCausalLM.from_pretrained(d, load_in_low_bit="sym_int8").half().to("cuda")
    eng = q.native_engine()
    start, recent, gen, cap = 4, 48, 8, 128
    eng.ensure_cache(1, cap, fp8=False)
    eng.cache.reset()
    dropped = 0
    with torch.no_grad():
        for turn in range(8):
            ids = torch.randint(3, 320, (1, 24), device="cuda")
            if eng.cache.host_len + 24 + gen > cap:
                dropped += eng.evict_middle(start, recent)
            logits = eng.prefill(ids)
            nxt = logits[:, -1].argmax(-1).view(1, 1)
            for _ in range(gen):
                lg, n2 = eng.decode(nxt)
                assert torch.isfinite(lg).all()
                nxt = n2.view(1, 1)
            assert eng.cache.host_len <= cap
    assert dropped > 0
