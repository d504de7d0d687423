"""Remote-code families on the native engine (GPU): engine logits / greedy tokens vs the plain PyTorch forward over the
same quantized weights."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _helpers():
    spec = importlib.util.spec_from_file_location("arch_cases", os.path.join(os.path.dirname(__file__), "test_arch_cpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.H, mod.D = 256, 64            # the attention kernels cover head_dim 64 / 128
    return mod


@pytest.mark.parametrize("case", ["chatglm3", "internlm2", "qwen", "minicpm", "internlm", "baichuan2_13b"])
def test_engine_matches_cpu_forward(case):
    from bigdl_amd.models import arch
    T = _helpers()
    torch.manual_seed(0)
    make, _ = T.CASES[case]
    cfg, sd = make()
    model = arch.NativeCausalLM.from_state_dict(cfg, sd, low_bit="sym_int8")
    ids = torch.randint(0, T.V, (1, 24))
    want, _ = model(ids)                                   # CPU: quantized matmul + plain attention
    ref_tokens = model.generate(ids, max_new_tokens=6)
    model.to("cuda")
    got, _ = model(ids.cuda())
    rel = ((got.float().cpu() - want).abs().mean() / want.abs().mean()).item()
    assert rel < 0.03, rel
    out = model.generate(ids.cuda(), max_new_tokens=6)
    assert (out.cpu() == ref_tokens).float().mean() > 0.9


LN_FAMILIES = ["gpt_neox", "gpt_neox_serial", "bloom", "bloom_6heads", "mpt_6heads", "falcon_7b", "falcon_40b", "falcon_rw",
               "gpt2", "gpt_bigcode", "opt", "phi", "starcoder2", "stablelm", "gptj",
               "minicpm3", "minicpm3_plain_q", "deepseek_v2", "deepseek_v2_grouped", "deepseek_v3"]        # latent attention (MLA), + routed / shared experts


@pytest.mark.parametrize("family", LN_FAMILIES)
def test_layernorm_family_engine_matches_cpu_forward(family):
    """LayerNorm / ALiBi / learned-position / parallel-residual families on the hipGraph engine against the CPU forward
    over the same quantized weights (which tests/test_arch_hf_cpu.py pins to HuggingFace's implementation)."""
    import json
    from bigdl_amd.models import arch
    spec = importlib.util.spec_from_file_location("hf_cases", os.path.join(os.path.dirname(__file__), "test_arch_hf_cpu.py"))
    H = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(H)
    H.SCALE = 4
    cfg, hf = H.FAMILIES[family]()
    cd = json.loads(cfg.to_json_string())
    sd = {k: v.detach() for k, v in hf.state_dict().items()}
    model = arch.NativeCausalLM.from_state_dict(cd, sd, low_bit="sym_int8")
    torch.manual_seed(3)
    ids = torch.randint(0, H.V, (1, 21))
    want, _ = model(ids)
    ref_tokens = model.generate(ids, max_new_tokens=6)
    model.to("cuda")
    got, _ = model(ids.cuda())
    rel = ((got.float().cpu() - want).abs().mean() / want.abs().mean()).item()
    assert rel < 0.03, rel
    out = model.generate(ids.cuda(), max_new_tokens=6)
    assert (out.cpu() == ref_tokens).float().mean() > 0.9
    # batched decode rows agree with the single-sequence run
    out2 = model.generate(ids.repeat(2, 1).cuda(), max_new_tokens=6)
    assert torch.equal(out2[0], out2[1])


def test_padded_head_dim_engine_matches_cpu_forward():
    """head_dim 80 (phi-2 class) runs on the engine as zero-padded 96-wide heads"""
    import json
    import transformers as T
    from bigdl_amd.models import arch
    torch.manual_seed(0)
    cfg = T.PhiConfig(vocab_size=96, hidden_size=320, num_hidden_layers=2, intermediate_size=512, num_attention_heads=4,
                      partial_rotary_factor=0.4, max_position_embeddings=128)
    hf = T.PhiForCausalLM(cfg).eval().float()
    cd = json.loads(cfg.to_json_string())
    model = arch.NativeCausalLM.from_state_dict(cd, {k: v.detach() for k, v in hf.state_dict().items()}, low_bit="sym_int8")
    assert model.dc.head_dim == 96
    ids = torch.randint(0, 96, (1, 21))
    want, _ = model(ids)
    ref_tokens = model.generate(ids, max_new_tokens=6)
    model.to("cuda")
    got, _ = model(ids.cuda())
    rel = ((got.float().cpu() - want).abs().mean() / want.abs().mean()).item()
    assert rel < 0.03, rel
    out = model.generate(ids.cuda(), max_new_tokens=6)
    assert (out.cpu() == ref_tokens).float().mean() > 0.9
