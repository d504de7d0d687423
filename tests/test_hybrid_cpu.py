"""CPU-GPU hybrid placement (experts on the host, the rest on the device)."""
import pytest
import torch
from transformers import MixtralConfig, MixtralForCausalLM

from bigdl_amd import optimize_model
from bigdl_amd.transformers.hybrid import HostResident, offload_to_cpu


def _tiny():
    torch.manual_seed(0)
    cfg = MixtralConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                        num_key_value_heads=2, num_local_experts=4, num_experts_per_tok=2, max_position_embeddings=128)
    return optimize_model(MixtralForCausalLM(cfg).eval(), low_bit="sym_int8")


def test_offload_wraps_experts_and_keeps_outputs():
    m = _tiny()
    ids = torch.randint(0, 128, (2, 7))
    want = m(ids).logits
    assert offload_to_cpu(m, skip_layers=[0]) == 2
    wrapped = [k for k, v in m.named_modules() if isinstance(v, HostResident)]
    assert wrapped == ["model.layers.1.mlp.experts", "model.layers.2.mlp.experts"]
    torch.testing.assert_close(m(ids).logits, want)
    assert offload_to_cpu(m) == 1            # layer 0 now; already wrapped ones are left alone
    # moving / casting the parent does not touch host-resident weights
    before = m.model.layers[1].mlp.experts.inner.gate_proj.weight.data.clone()
    m.to(torch.float32)
    assert torch.equal(m.model.layers[1].mlp.experts.inner.gate_proj.weight.data, before)
    from bigdl_amd.models.hf_patch import native_engine_supported
    assert not native_engine_supported(m)


@pytest.mark.gpu
def test_hybrid_model_on_gpu_matches_all_gpu():
    m = _tiny()
    ids = torch.randint(0, 128, (1, 9))
    full = m.to("cuda")
    want = full(ids.cuda()).logits.float().cpu()
    h = _tiny()
    offload_to_cpu(h)
    h.to("cuda")
    assert h.model.layers[0].mlp.experts.inner.gate_proj.weight.device.type == "cpu"
    assert h.model.layers[0].self_attn.qkv_proj.weight.device.type == "cuda"
    got = h(ids.cuda()).logits.float().cpu()
    assert ((got - want).abs().mean() / want.abs().mean()).item() < 0.02
    out = h.generate(ids.cuda(), max_new_tokens=4, do_sample=False)
    assert out.shape == (1, 13)
