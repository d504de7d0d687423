"""Breadth check of the module path: tiny random-init models of the HF families the reference lists
(python/llm/src/ipex_llm/transformers/convert.py:1301-2025) go through ``optimize_model`` and keep their logits.

Criterion is the reference's own (python/llm/test/inference/test_transformers_api.py:151-169): mean relative
difference between optimized and original logits below 5 %.  sym_int8 keeps quantization noise out of the way so
that a wrong rewrite (merged qkv order, transposed Conv1D, replaced router) shows up clearly."""
import copy

import pytest
import torch
from transformers import AutoConfig, AutoModelForCausalLM

from bigdl_amd.optimize import optimize_model

COMMON = dict(vocab_size=320, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=256)
FAMILIES = {
    "llama": dict(intermediate_size=256, num_key_value_heads=2),
    "mistral": dict(intermediate_size=256, num_key_value_heads=2),
    "qwen2": dict(intermediate_size=256, num_key_value_heads=2),
    "qwen3": dict(intermediate_size=256, num_key_value_heads=2, head_dim=32),
    "gemma": dict(intermediate_size=256, num_key_value_heads=2, head_dim=32),
    "gemma2": dict(intermediate_size=256, num_key_value_heads=2, head_dim=32),
    "phi": dict(intermediate_size=256),
    "phi3": dict(intermediate_size=256, num_key_value_heads=2, pad_token_id=0),
    "stablelm": dict(intermediate_size=256, num_key_value_heads=2),
    "starcoder2": dict(intermediate_size=256, num_key_value_heads=2),
    "gpt2": dict(n_embd=128, n_layer=2, n_head=4, n_positions=256),
    "gpt_neox": dict(intermediate_size=256),
    "gpt_bigcode": dict(n_embd=128, n_layer=2, n_head=4, n_positions=256),
    "gptj": dict(n_embd=128, n_layer=2, n_head=4, n_positions=256, rotary_dim=16),
    "bloom": dict(n_head=4, n_layer=2),
    "mpt": dict(d_model=128, n_heads=4, n_layers=2, max_seq_len=256),
    "falcon": dict(),
    "glm": dict(intermediate_size=256, num_key_value_heads=2, head_dim=32, pad_token_id=0),
    "glm4": dict(intermediate_size=256, num_key_value_heads=2, head_dim=32, pad_token_id=0),
    "minicpm3": dict(intermediate_size=256, num_key_value_heads=4, q_lora_rank=32, kv_lora_rank=16, qk_nope_head_dim=16,
                     qk_rope_head_dim=16, v_head_dim=16),
    "cohere": dict(intermediate_size=256),
    "granite": dict(intermediate_size=256, num_key_value_heads=2),
    "phimoe": dict(intermediate_size=128, num_key_value_heads=2, num_local_experts=4),
    "deepseek_v2": dict(intermediate_size=256, moe_intermediate_size=64, num_key_value_heads=4, n_routed_experts=4,
                        n_shared_experts=1, q_lora_rank=32, kv_lora_rank=16, qk_nope_head_dim=16, qk_rope_head_dim=16,
                        v_head_dim=16, first_k_dense_replace=1, num_experts_per_tok=2),
    "deepseek_v3": dict(intermediate_size=256, moe_intermediate_size=64, num_key_value_heads=4, n_routed_experts=4,
                        n_shared_experts=1, q_lora_rank=32, kv_lora_rank=16, qk_nope_head_dim=16, qk_rope_head_dim=16,
                        v_head_dim=16, first_k_dense_replace=1, num_experts_per_tok=2, n_group=2, topk_group=1),
    "qwen3_moe": dict(intermediate_size=256, moe_intermediate_size=64, num_key_value_heads=2, head_dim=32, num_experts=4,
                      num_experts_per_tok=2),
    "glm4_moe": dict(intermediate_size=256, moe_intermediate_size=64, num_key_value_heads=2, head_dim=32, n_routed_experts=4,
                     n_shared_experts=1, num_experts_per_tok=2, first_k_dense_replace=1, n_group=1, topk_group=1,
                     pad_token_id=0),
    # (experts with biases and a clamped activation of their own: they stay dense, attention is converted)
    "gpt_oss": dict(intermediate_size=128, num_key_value_heads=2, head_dim=32, num_local_experts=4, num_experts_per_tok=2,
                    sliding_window=64, layer_types=["sliding_attention", "full_attention"]),
    "olmo2": dict(intermediate_size=256, num_key_value_heads=2),
    "olmoe": dict(intermediate_size=128, num_key_value_heads=4, num_experts=4, num_experts_per_tok=2),
    "mixtral": dict(intermediate_size=128, num_key_value_heads=2, num_local_experts=4),
    "qwen2_moe": dict(intermediate_size=256, moe_intermediate_size=64, shared_expert_intermediate_size=128,
                      num_key_value_heads=2, num_experts=4, num_experts_per_tok=2),
    "exaone4": dict(intermediate_size=256, num_key_value_heads=2, head_dim=32),
    "gemma3_text": dict(intermediate_size=256, num_key_value_heads=2, head_dim=32),
    "nemotron": dict(intermediate_size=256, num_key_value_heads=2),
    "codegen": dict(n_embd=128, n_layer=2, n_head=4, n_positions=256, rotary_dim=16),
    "xglm": dict(d_model=128, num_layers=2, attention_heads=4, ffn_dim=256),
}


def _rel(a, b):
    return ((a - b).abs().mean() / b.abs().mean()).item()


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_causal_lm_family_keeps_logits(family):
    torch.manual_seed(0)
    cfg = AutoConfig.for_model(family, **{**COMMON, **FAMILIES[family]})
    model = AutoModelForCausalLM.from_config(cfg).eval().float()
    ids = torch.randint(0, 320, (2, 9))        # <= 10 positions: the lm_head keeps every position
    with torch.no_grad():
        ref = model(ids).logits.float()
        q = optimize_model(copy.deepcopy(model), low_bit="sym_int8")
        out = q(ids).logits.float()
        gen = q.generate(ids[:1], max_new_tokens=3, do_sample=False)
    n_low = sum(1 for m in q.modules() if type(m).__name__ in ("LowBitLinear", "LowBitExperts"))
    assert n_low >= 2 * getattr(cfg, "num_hidden_layers", 2), "projections were not converted"
    assert _rel(out, ref) < 0.05
    assert gen.shape == (1, 12)


def test_bert_encoder_and_whisper_seq2seq():
    from transformers import BertConfig, BertModel, WhisperConfig, WhisperForConditionalGeneration
    torch.manual_seed(0)
    bert = BertModel(BertConfig(vocab_size=320, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                                intermediate_size=256, max_position_embeddings=64)).eval()
    ids = torch.randint(0, 320, (2, 16))
    with torch.no_grad():
        ref = bert(ids).last_hidden_state
        out = optimize_model(copy.deepcopy(bert), low_bit="sym_int8")(ids).last_hidden_state
    assert _rel(out, ref) < 0.05
    wcfg = WhisperConfig(vocab_size=320, d_model=128, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                         decoder_attention_heads=4, encoder_ffn_dim=256, decoder_ffn_dim=256, num_mel_bins=16,
                         max_source_positions=32, max_target_positions=32, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                         decoder_start_token_id=1)
    wh = WhisperForConditionalGeneration(wcfg).eval()
    feats = torch.randn(1, 16, 64)
    dec = torch.randint(3, 320, (1, 6))
    with torch.no_grad():
        ref = wh(input_features=feats, decoder_input_ids=dec).logits
        q = optimize_model(copy.deepcopy(wh), low_bit="sym_int8")
        out = q(input_features=feats, decoder_input_ids=dec).logits
    assert _rel(out, ref) < 0.05


def test_vision_language_model_converts_every_tower():
    """LLaVA-style wrapper (CLIP vision tower + projector + llama): the conversion reaches the language model through the
    wrapper, quantizes tower and projector linears too, and image + text prompts give the logits of the fp32 model."""
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear
    torch.manual_seed(0)
    vc = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=28,
                          patch_size=14, projection_dim=64)
    tc = LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                     num_key_value_heads=2, max_position_embeddings=256)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_index=319, vision_feature_layer=-1,
                      vision_feature_select_strategy="full")
    m = LlavaForConditionalGeneration(cfg).eval()
    n_img = (28 // 14) ** 2 + 1
    ids = torch.cat([torch.tensor([[1, 5]]), torch.full((1, n_img), 319), torch.randint(2, 300, (1, 6))], 1)
    px = torch.randn(1, 3, 28, 28)
    with torch.no_grad():
        want = m(input_ids=ids, pixel_values=px).logits[:, -1]
    for low_bit, tol in (("sym_int8", 0.03), ("fp16", 0.01)):
        import copy
        q = optimize_model(copy.deepcopy(m), low_bit=low_bit)
        parts = {".".join(n.split(".")[:2]) for n, mod in q.named_modules() if isinstance(mod, LowBitLinear) or
                 type(mod).__name__ == "FP16Linear"}
        assert {"lm_head", "model.language_model", "model.multi_modal_projector", "model.vision_tower"} <= parts, parts
        with torch.no_grad():
            got = q(input_ids=ids, pixel_values=px).logits[:, -1].float()      # the lm_head keeps the last position only
        assert _rel(got, want) < tol, (low_bit, _rel(got, want))
        out = q.generate(input_ids=ids, pixel_values=px, max_new_tokens=4, do_sample=False)
        assert out.shape == (1, ids.shape[1] + 4)


# ---- multimodal wrappers and vision towers (reference: models/{internvl,chatglm4v,minicpmv}.py and the nested-model
# handling of convert.py:1301-1400): tiny random-init models keep their logits after optimize_model ----
_TEXT = dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
             num_key_value_heads=2, max_position_embeddings=256)


def _llava():
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig
    cfg = LlavaConfig(vision_config=CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                                     image_size=28, patch_size=14),
                      text_config=LlamaConfig(**_TEXT), image_token_index=319, vision_feature_layer=-1,
                      vision_feature_select_strategy="full")
    n_img = (28 // 14) ** 2 + 1
    return cfg, dict(input_ids=torch.tensor([[1, 5] + [319] * n_img + [7, 8, 9]]), pixel_values=torch.randn(1, 3, 28, 28))


def _internvl():
    from transformers import InternVLConfig
    cfg = InternVLConfig(vision_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                            image_size=[28, 28], patch_size=[14, 14]),
                         text_config=dict(model_type="qwen2", **_TEXT), image_token_id=319, downsample_ratio=0.5)
    return cfg, dict(input_ids=torch.tensor([[1, 5, 319, 7, 8, 9]]), pixel_values=torch.randn(1, 3, 28, 28))


def _glm4v():
    from transformers import Glm4vConfig
    cfg = Glm4vConfig(text_config=dict(rope_parameters={"rope_type": "default", "mrope_section": [4, 6, 6], "rope_theta": 10000.0},
                                       **_TEXT),
                      vision_config=dict(hidden_size=64, depth=2, num_heads=4, intermediate_size=128, out_hidden_size=128,
                                         patch_size=14, image_size=28, spatial_merge_size=2, temporal_patch_size=2),
                      image_token_id=319, video_token_id=318, image_start_token_id=317, image_end_token_id=316)
    return cfg, dict(input_ids=torch.tensor([[1, 5, 7, 8, 9, 10]]))      # text-only turn through the 3-axis rotary path


@pytest.mark.parametrize("build", [_llava, _internvl, _glm4v], ids=["llava", "internvl", "glm4v"])
def test_multimodal_wrapper_keeps_logits(build):
    from transformers import AutoModelForImageTextToText
    torch.manual_seed(0)
    cfg, inputs = build()
    model = AutoModelForImageTextToText.from_config(cfg).eval().float()
    with torch.no_grad():
        ref = model(**inputs).logits.float()
        q = optimize_model(copy.deepcopy(model), low_bit="sym_int8")
        out = q(**inputs).logits.float()
    assert sum(1 for m in q.modules() if type(m).__name__ == "LowBitLinear") >= 8      # text and vision projections
    assert _rel(out, ref) < 0.05


def test_siglip_vision_tower_with_72_wide_heads():
    """MiniCPM-V's image encoder shape (reference models/minicpmv.py:44-90: heads of 72 channels)"""
    from transformers import SiglipVisionConfig, SiglipVisionModel
    torch.manual_seed(0)
    m = SiglipVisionModel(SiglipVisionConfig(hidden_size=144, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                             image_size=28, patch_size=14)).eval().float()
    x = torch.randn(2, 3, 28, 28)
    with torch.no_grad():
        ref = m(pixel_values=x).last_hidden_state
        out = optimize_model(copy.deepcopy(m), low_bit="sym_int8")(pixel_values=x).last_hidden_state
    assert _rel(out, ref) < 0.05


def test_idefics2_and_qwen2_audio_wrappers_keep_logits():
    """Idefics2 (SigLIP tower + perceiver resampler + mistral) and Qwen2-Audio (whisper-style encoder + qwen2): the other
    two multimodal families HuggingFace ships that the reference converts (convert.py: "idefics2", "qwen2_audio")"""
    from transformers import (Idefics2Config, Idefics2ForConditionalGeneration, Qwen2AudioConfig,
                              Qwen2AudioForConditionalGeneration)
    torch.manual_seed(0)
    cfg = Idefics2Config(vision_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                            image_size=28, patch_size=14),
                         perceiver_config=dict(hidden_size=128, resampler_n_latents=4, resampler_depth=1, resampler_n_heads=4,
                                               resampler_head_dim=32, num_key_value_heads=2),
                         text_config=dict(model_type="mistral", **_TEXT), image_token_id=319)
    m = Idefics2ForConditionalGeneration(cfg).eval().float()
    ids, px = torch.tensor([[1, 5] + [319] * 4 + [7, 8, 9]]), torch.randn(1, 1, 3, 28, 28)
    with torch.no_grad():
        ref = m(input_ids=ids, pixel_values=px).logits.float()
        q = optimize_model(copy.deepcopy(m), low_bit="sym_int8")
        out = q(input_ids=ids, pixel_values=px).logits.float()
    assert sum(1 for x in q.modules() if type(x).__name__ == "LowBitLinear") >= 20
    assert _rel(out, ref) < 0.05
    cfg = Qwen2AudioConfig(audio_config=dict(d_model=64, encoder_layers=2, encoder_attention_heads=4, encoder_ffn_dim=128,
                                             num_mel_bins=16, max_source_positions=16),
                           text_config=dict(model_type="qwen2", **_TEXT), audio_token_index=319)
    m = Qwen2AudioForConditionalGeneration(cfg).eval().float()
    ids = torch.tensor([[1, 5, 7, 8, 9]])
    with torch.no_grad():
        ref = m(input_ids=ids).logits.float()
        out = optimize_model(copy.deepcopy(m), low_bit="sym_int8")(input_ids=ids).logits.float()
    assert _rel(out, ref) < 0.05


def test_t5_encoder_decoder_through_the_seq2seq_auto_class(tmp_path):
    """flan-t5 (reference example/GPU/HuggingFace/LLM/flan-t5/generate.py:47-52): AutoModelForSeq2SeqLM with
    ``modules_to_not_convert=["wo", "lm_head"]``; logits and generation stay those of the fp32 model."""
    from transformers import T5Config, T5ForConditionalGeneration
    from bigdl_amd.transformers import AutoModelForSeq2SeqLM
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear
    torch.manual_seed(0)
    cfg = T5Config(vocab_size=320, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=4,
                   feed_forward_proj="gated-gelu", decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
                   tie_word_embeddings=False)
    m = T5ForConditionalGeneration(cfg).eval()
    m.save_pretrained(tmp_path)
    ids = torch.randint(3, 320, (2, 12))
    dec = torch.randint(3, 320, (2, 5))
    with torch.no_grad():
        ref = m(input_ids=ids, decoder_input_ids=dec).logits
        q = AutoModelForSeq2SeqLM.from_pretrained(str(tmp_path), load_in_low_bit="sym_int8", optimize_model=False,
                                                  modules_to_not_convert=["wo", "lm_head"])
        out = q(input_ids=ids, decoder_input_ids=dec).logits
        gen = q.generate(ids, max_new_tokens=6, do_sample=False, use_cache=True)
        gen_ref = m.generate(ids, max_new_tokens=6, do_sample=False, use_cache=True)
    blk = q.encoder.block[0].layer
    assert isinstance(blk[0].SelfAttention.q, LowBitLinear) and not isinstance(blk[1].DenseReluDense.wo, LowBitLinear)
    assert not isinstance(q.lm_head, LowBitLinear)
    assert _rel(out, ref) < 0.05
    assert gen.shape == gen_ref.shape and (gen == gen_ref).float().mean() > 0.7
    # everything converted (``wo`` and the head too), optimizations on: the feed-forward block no longer casts its hidden
    # states to the byte type of a packed ``wo``, the head scores every position of a teacher-forced pass
    q2 = optimize_model(copy.deepcopy(m), low_bit="sym_int8")
    assert isinstance(q2.encoder.block[0].layer[1].DenseReluDense.wo, LowBitLinear) and isinstance(q2.lm_head, LowBitLinear)
    with torch.no_grad():
        assert _rel(q2(input_ids=ids, decoder_input_ids=dec).logits, ref) < 0.05
        gen2 = q2.generate(ids, max_new_tokens=6, do_sample=False, use_cache=True)
    assert gen2.shape == gen_ref.shape and (gen2 == gen_ref).float().mean() > 0.7


@pytest.mark.parametrize("model_type", ["mamba", "falcon_mamba", "bitnet", "bigbird_pegasus", "llama4_text", "inkling_text",
                                        "opt", "gpt_neo", "persimmon", "xlstm"])
def test_generic_patch_leaves_unlisted_families_intact(model_type):
    """Families nobody wrote a rule for go through the same generic passes; each of these once broke one of them: a head
    whose input is cast to ``lm_head.weight.dtype`` (mamba), an RMSNorm whose weight is never applied (falcon_mamba),
    a gated MLP with a norm inside (bitnet) or extra arguments (inkling), a model that goes through the attention
    registry but is only right in eager form (bigbird_pegasus), fused experts in transposed layout (llama4)."""
    import inspect
    from transformers import AutoModelForCausalLM
    from transformers.models.auto.configuration_auto import CONFIG_MAPPING
    if model_type not in CONFIG_MAPPING:
        pytest.skip("not in this transformers version")
    common = dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                  num_key_value_heads=2, max_position_embeddings=128, n_embd=128, n_layer=2, n_head=4, n_positions=128,
                  d_model=128, ffn_dim=256, n_inner=256, head_dim=32, num_experts=4, num_local_experts=4,
                  num_experts_per_tok=2, moe_intermediate_size=64, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    cfg_cls = CONFIG_MAPPING[model_type]
    cfg = cfg_cls(**{k: v for k, v in common.items() if k in inspect.signature(cfg_cls.__init__).parameters})
    torch.manual_seed(0)
    m = AutoModelForCausalLM.from_config(cfg).eval()
    ids = torch.randint(3, 320, (2, 10))
    with torch.no_grad():
        ref = m(ids).logits
        out = optimize_model(copy.deepcopy(m), low_bit="sym_int8")(ids).logits
    assert out.shape == ref.shape and _rel(out, ref) < 0.08
