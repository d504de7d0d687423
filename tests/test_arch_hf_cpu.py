"""LayerNorm families: the weight-name maps + the plain forward reproduce HuggingFace's own implementation of each
family on tiny random checkpoints (fp32, CPU)."""
import json

import pytest
import torch
import transformers as T

from bigdl_amd.models import arch

V = 96


SCALE = 1      # the GPU test widens every family 4x (head_dim 64, which the attention kernels cover)
_WIDTHS = ("hidden_size", "n_embd", "d_model", "intermediate_size", "ffn_dim", "word_embed_proj_dim", "rotary_dim",
           "moe_intermediate_size")


def _mk(cfg_cls, model_cls, **kw):
    torch.manual_seed(0)
    kw = {k: v * SCALE if k in _WIDTHS else v for k, v in kw.items()}
    cfg = cfg_cls(**kw)
    cfg._attn_implementation = "eager"
    model = model_cls(cfg).eval().float()
    # random biases / norm weights so that a missed bias or a swapped norm shows
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.copy_(torch.randn_like(p) * 0.1 + (1.0 if "norm" in n.lower() or "ln" in n.lower() else 0.0))
    return cfg, model


FAMILIES = {
    "gpt_neox": lambda: _mk(T.GPTNeoXConfig, T.GPTNeoXForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                            num_attention_heads=4, intermediate_size=128, rotary_pct=0.25, max_position_embeddings=64),
    "gpt_neox_serial": lambda: _mk(T.GPTNeoXConfig, T.GPTNeoXForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                                   num_attention_heads=4, intermediate_size=128, rotary_pct=0.5, use_parallel_residual=False,
                                   max_position_embeddings=64),
    "bloom": lambda: _mk(T.BloomConfig, T.BloomForCausalLM, vocab_size=V, hidden_size=64, n_layer=2, n_head=4),
    "bloom_6heads": lambda: _mk(T.BloomConfig, T.BloomForCausalLM, vocab_size=V, hidden_size=96, n_layer=2, n_head=6),
    "mpt": lambda: _mk(T.MptConfig, T.MptForCausalLM, vocab_size=V, d_model=64, n_heads=4, n_layers=2, max_seq_len=64),
    "mpt_6heads": lambda: _mk(T.MptConfig, T.MptForCausalLM, vocab_size=V, d_model=96, n_heads=6, n_layers=2, max_seq_len=64),
    "falcon_7b": lambda: _mk(T.FalconConfig, T.FalconForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                             num_attention_heads=4, multi_query=True, parallel_attn=True, new_decoder_architecture=False,
                             bias=False, alibi=False),
    "falcon_40b": lambda: _mk(T.FalconConfig, T.FalconForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                              num_attention_heads=4, num_kv_heads=2, new_decoder_architecture=True, bias=False, alibi=False),
    "falcon_rw": lambda: _mk(T.FalconConfig, T.FalconForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                             num_attention_heads=4, multi_query=False, parallel_attn=False, new_decoder_architecture=False,
                             bias=True, alibi=True),
    "gpt2": lambda: _mk(T.GPT2Config, T.GPT2LMHeadModel, vocab_size=V, n_embd=64, n_layer=2, n_head=4, n_positions=64),
    "gpt_bigcode": lambda: _mk(T.GPTBigCodeConfig, T.GPTBigCodeForCausalLM, vocab_size=V, n_embd=64, n_layer=2, n_head=4,
                               n_positions=64, multi_query=True),
    "gpt_bigcode_mha": lambda: _mk(T.GPTBigCodeConfig, T.GPTBigCodeForCausalLM, vocab_size=V, n_embd=64, n_layer=2, n_head=4,
                                   n_positions=64, multi_query=False),
    "opt": lambda: _mk(T.OPTConfig, T.OPTForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2, ffn_dim=128,
                       num_attention_heads=4, max_position_embeddings=64, word_embed_proj_dim=64),
    "phi": lambda: _mk(T.PhiConfig, T.PhiForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2, intermediate_size=128,
                       num_attention_heads=4, partial_rotary_factor=0.5, max_position_embeddings=64),
    "starcoder2": lambda: _mk(T.Starcoder2Config, T.Starcoder2ForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                              intermediate_size=128, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=64,
                              sliding_window=None),
    "stablelm": lambda: _mk(T.StableLmConfig, T.StableLmForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                            intermediate_size=128, num_attention_heads=4, num_key_value_heads=2, partial_rotary_factor=0.25,
                            max_position_embeddings=64, use_qkv_bias=True),
    # multi-head latent attention (64 + 32 = 96 channels per head as in MiniCPM3-4B; without the q latent too)
    "minicpm3": lambda: _mk(T.MiniCPM3Config, T.MiniCPM3ForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=2,
                            intermediate_size=128, num_attention_heads=4, num_key_value_heads=4, q_lora_rank=64,
                            kv_lora_rank=64, qk_nope_head_dim=64, qk_rope_head_dim=32, v_head_dim=64,
                            max_position_embeddings=64, scale_emb=3.0, scale_depth=1.4, dim_model_base=32,
                            tie_word_embeddings=False),
    "minicpm3_plain_q": lambda: _mk(T.MiniCPM3Config, T.MiniCPM3ForCausalLM, vocab_size=V, hidden_size=64,
                                    num_hidden_layers=2, intermediate_size=128, num_attention_heads=4,
                                    num_key_value_heads=4, q_lora_rank=None, kv_lora_rank=64, qk_nope_head_dim=32,
                                    qk_rope_head_dim=16, v_head_dim=48, max_position_embeddings=64,
                                    tie_word_embeddings=False),
    # latent attention with interleaved rotary pairs + routed / shared experts after one dense layer
    "deepseek_v2": lambda: _mk(T.DeepseekV2Config, T.DeepseekV2ForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=3,
                               intermediate_size=128, moe_intermediate_size=64, num_attention_heads=4, num_key_value_heads=4,
                               q_lora_rank=None, kv_lora_rank=64, qk_nope_head_dim=32, qk_rope_head_dim=32, v_head_dim=48,
                               n_routed_experts=8, num_experts_per_tok=3, n_shared_experts=2, first_k_dense_replace=1,
                               routed_scaling_factor=1.5, max_position_embeddings=64, tie_word_embeddings=False),
    "deepseek_v2_grouped": lambda: _mk(T.DeepseekV2Config, T.DeepseekV2ForCausalLM, vocab_size=V, hidden_size=64,
                                       num_hidden_layers=2, intermediate_size=128, moe_intermediate_size=64,
                                       num_attention_heads=4, num_key_value_heads=4, q_lora_rank=64, kv_lora_rank=64,
                                       qk_nope_head_dim=32, qk_rope_head_dim=32, v_head_dim=32, n_routed_experts=16,
                                       num_experts_per_tok=4, n_shared_experts=1, first_k_dense_replace=0,
                                       topk_method="group_limited_greedy", n_group=4, topk_group=2,
                                       routed_scaling_factor=2.0, max_position_embeddings=64, tie_word_embeddings=False),
    # sigmoid router scores, learned selection bias, groups ranked by their two best experts, renormalized weights
    "deepseek_v3": lambda: _mk_v3(),
    "gptj": lambda: _mk(T.GPTJConfig, T.GPTJForCausalLM, vocab_size=V, n_embd=64, n_layer=2, n_head=4, rotary_dim=8,
                        n_positions=64),
}


def _mk_v3():
    cfg, model = _mk(T.DeepseekV3Config, T.DeepseekV3ForCausalLM, vocab_size=V, hidden_size=64, num_hidden_layers=3,
                     intermediate_size=128, moe_intermediate_size=64, num_attention_heads=4, num_key_value_heads=4,
                     q_lora_rank=64, kv_lora_rank=64, qk_nope_head_dim=32, qk_rope_head_dim=32, v_head_dim=32,
                     n_routed_experts=16, num_experts_per_tok=4, n_shared_experts=1, first_k_dense_replace=1, n_group=4,
                     topk_group=2, routed_scaling_factor=2.5, norm_topk_prob=True, max_position_embeddings=64,
                     tie_word_embeddings=False)
    with torch.no_grad():
        for m in model.modules():                       # the correction bias is zero at initialisation: make it matter
            if hasattr(m, "e_score_correction_bias"):
                m.e_score_correction_bias.copy_(torch.randn_like(m.e_score_correction_bias) * 0.2)
    return cfg, model


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_weight_map_reproduces_huggingface_forward(family):
    cfg, model = FAMILIES[family]()
    ids = torch.randint(0, V, (2, 13))
    with torch.no_grad():
        want = model(ids).logits.float()
    cd = json.loads(cfg.to_json_string())
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    dc, canon = arch.ARCH_CONVERTERS[cd["model_type"]](cd, sd)
    plain = dc
    if family == "falcon_rw":
        # this transformers version adds the ALiBi bias twice on its only code path for alibi models (once folded into
        # the mask, once next to the scores); the converter keeps the single bias of the original implementation, so
        # the comparison runs with doubled slopes
        import dataclasses
        dc = dataclasses.replace(dc, alibi_slopes=tuple(2 * x for x in dc.alibi_slopes))
    got, past = arch.reference_forward(dc, canon, ids)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    step, _ = arch.reference_forward(dc, canon, ids[:, -1:], [(k[:, :, :-1], v[:, :, :-1]) for k, v in past])
    torch.testing.assert_close(step[:, 0], want[:, -1], rtol=2e-4, atol=2e-4)
    # the config-only path (used by load_low_bit) rebuilds the same DecoderConfig
    assert arch.arch_config_only(cd) == plain


def test_from_pretrained_native_engine_opt_in(tmp_path):
    """A family HuggingFace has a class for goes to the native weight map only when asked to."""
    from ipex_llm.transformers import AutoModelForCausalLM
    cfg, model = FAMILIES["gpt_neox"]()
    model.save_pretrained(str(tmp_path / "ckpt"), safe_serialization=True)
    ids = torch.randint(0, V, (1, 9))
    with torch.no_grad():
        want = model(ids).logits
    hf_path = AutoModelForCausalLM.from_pretrained(str(tmp_path / "ckpt"), load_in_low_bit="sym_int8")
    assert not isinstance(hf_path, arch.NativeCausalLM)
    native = AutoModelForCausalLM.from_pretrained(str(tmp_path / "ckpt"), load_in_low_bit="sym_int8", native_engine=True)
    assert isinstance(native, arch.NativeCausalLM) and native.dc.norm == "layer" and native.dc.parallel_residual
    got, _ = native(ids)
    assert ((got - want).abs().mean() / want.abs().mean()).item() < 0.05
    native.save_low_bit(str(tmp_path / "lb"))
    again = AutoModelForCausalLM.load_low_bit(str(tmp_path / "lb"))
    assert again.dc == native.dc
    assert torch.equal(again.generate(ids, max_new_tokens=3), native.generate(ids, max_new_tokens=3))


def test_odd_head_dim_is_padded_to_a_kernel_size():
    """phi-2 class head_dim 80 -> 96 by zero-padding heads: same logits, original softmax scale and rotary dim."""
    torch.manual_seed(0)
    cfg, model = _mk(T.PhiConfig, T.PhiForCausalLM, vocab_size=V, hidden_size=320, num_hidden_layers=2, intermediate_size=256,
                     num_attention_heads=4, partial_rotary_factor=0.4, max_position_embeddings=64)
    ids = torch.randint(0, V, (2, 12))
    with torch.no_grad():
        want = model(ids).logits.float()
    cd = json.loads(cfg.to_json_string())
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    dc, canon = arch.ARCH_CONVERTERS["phi"](cd, sd)
    assert dc.head_dim == 80
    pdc, pcanon = arch.pad_head_dim(dc, canon)
    assert pdc.head_dim == 96 and pdc.rot_dim == 32 and abs(pdc.attn_scale - 80 ** -0.5) < 1e-9
    assert pcanon["L0.qkv"].shape[0] == 3 * 4 * 96 and pcanon["L0.o"].shape[1] == 4 * 96
    got, _ = arch.reference_forward(pdc, pcanon, ids)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    assert arch.arch_config_only(cd, padded=True) == pdc


def test_latent_attention_moe_checkpoint_round_trip(tmp_path):
    """deepseek_v2 through the public API with the native map: low-bit conversion tracks the fp32 model, and the saved
    low-bit checkpoint (latent projections, router, stacked experts, shared experts) loads back bit-identically."""
    from ipex_llm.transformers import AutoModelForCausalLM
    cfg, model = FAMILIES["deepseek_v2"]()
    model.save_pretrained(str(tmp_path / "ckpt"), safe_serialization=True)
    ids = torch.randint(0, V, (1, 9))
    with torch.no_grad():
        want = model(ids).logits
    native = AutoModelForCausalLM.from_pretrained(str(tmp_path / "ckpt"), load_in_low_bit="sym_int8", native_engine=True)
    assert isinstance(native, arch.NativeCausalLM) and native.dc.mla_kv_lora == 64 and native.dc.num_experts == 8
    got, _ = native(ids)
    assert ((got - want).abs().mean() / want.abs().mean()).item() < 0.05
    native.save_low_bit(str(tmp_path / "lb"))
    again = AutoModelForCausalLM.load_low_bit(str(tmp_path / "lb"))
    assert again.dc == native.dc
    assert torch.equal(again.generate(ids, max_new_tokens=3), native.generate(ids, max_new_tokens=3))
