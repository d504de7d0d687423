"""QLoRA on the GPU: frozen low-bit base (device-tiled weights, dequant + bf16 GEMM forward, grad @ dequant(W)
backward), bf16 adapters, ReLoRA merge with re-quantization on the device."""
import pytest
import torch
from transformers import LlamaConfig, LlamaForCausalLM

pytestmark = pytest.mark.gpu


def _qmodel(low_bit):
    from bigdl_amd.optimize import optimize_model
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128, tie_word_embeddings=False)
    m = optimize_model(LlamaForCausalLM(cfg), low_bit=low_bit, optimize_llm=False, modules_to_not_convert=["lm_head"])
    return m.to("cuda")


@pytest.mark.parametrize("mode,low_bit", [("qlora", "nf4"), ("qalora", "sym_int4")])
def test_qlora_training_on_gpu(mode, low_bit):
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear, dequant_weight
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model, prepare_model_for_kbit_training
    model = prepare_model_for_kbit_training(_qmodel(low_bit), use_gradient_checkpointing=True)
    pm = get_peft_model(model, LoraConfig(r=8, lora_alpha=16, target_modules=["q_proj", "k_proj", "v_proj", "o_proj",
                                                                              "up_proj", "down_proj", "gate_proj"],
                                          lora_dropout=0.0, training_mode=mode))
    base = pm.base_model.model.layers[0].self_attn.q_proj.base_layer
    assert isinstance(base, LowBitLinear) and base.weight.qlayout == "t16" and base.compute_dtype == torch.bfloat16
    before = base.weight.data.clone()
    ids = torch.randint(0, 320, (4, 32), device="cuda")
    opt = torch.optim.AdamW([p for p in pm.parameters() if p.requires_grad], lr=5e-3)
    pm.train()
    losses = []
    for _ in range(12):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = pm(input_ids=ids, labels=ids).loss
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 0.2, losses
    assert torch.equal(base.weight.data, before)                    # the quantized base never changes
    assert all(p.grad is None for n, p in pm.named_parameters() if "lora_" not in n)
    # activation gradient of the frozen matmul = grad_out @ dequant(W)
    x = torch.randn(5, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    base.train()
    y = base(x)
    g = torch.randn_like(y)
    y.backward(g)
    ref = g.float() @ dequant_weight(base.weight, torch.float32)
    torch.testing.assert_close(x.grad.float(), ref, rtol=3e-2, atol=3e-2)


def test_relora_merge_requantizes_on_gpu():
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model
    from bigdl_amd.transformers.relora import merge_and_reinit
    pm = get_peft_model(_qmodel("sym_int8"), LoraConfig(r=4, lora_alpha=8, target_modules=["q_proj", "v_proj"],
                                                        training_mode="relora"))
    ids = torch.randint(0, 320, (2, 16), device="cuda")
    opt = torch.optim.AdamW([p for p in pm.parameters() if p.requires_grad], lr=1e-2)
    for _ in range(3):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pm(input_ids=ids, labels=ids).loss.backward()
        opt.step()
        opt.zero_grad()
    pm.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        before = pm(input_ids=ids).logits.float()
    merge_and_reinit(pm)
    base = pm.base_model.model.layers[0].self_attn.q_proj.base_layer
    assert base.weight.data.device.type == "cuda" and base.weight.qlayout == "t16"
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        after = pm(input_ids=ids).logits.float()
    assert ((after - before).abs().mean() / before.abs().mean()).item() < 0.03


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_fused_lora_node_matches_module_by_module(monkeypatch, p):
    """LoraLowBitLinear's one-node GPU path (FusedLoraLowBit) against the module-by-module form: output, activation
    gradient and adapter gradients (dropout: same random stream, so the same mask)."""
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model, LoraLowBitLinear, FusedLoraLowBit
    pm = get_peft_model(_qmodel("nf4"), LoraConfig(r=8, lora_alpha=16, target_modules=["q_proj"], lora_dropout=p,
                                                   training_mode="qlora"))
    lin = pm.base_model.model.layers[0].self_attn.q_proj
    assert isinstance(lin, LoraLowBitLinear)
    torch.nn.init.normal_(lin.lora_B["default"].weight, std=0.05)            # (zero-initialised: delta would vanish)
    lin.train()
    x0 = torch.randn(3, 40, 256, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(3, 40, lin.out_features, device="cuda", dtype=torch.bfloat16)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("BIGDL_FUSED_LORA", fused)
        x = x0.clone().requires_grad_(True)
        for prm in (lin.lora_A["default"].weight, lin.lora_B["default"].weight):
            prm.grad = None
        torch.manual_seed(123)
        assert lin._fusable(x) == (fused == "1")
        y = lin(x)
        y.backward(g)
        res[fused] = (y.detach().float(), x.grad.float(), lin.lora_A["default"].weight.grad.float().clone(),
                      lin.lora_B["default"].weight.grad.float().clone())
    if p > 0.0:
        # the two forms draw their dropout masks from different kernels: compare statistics only
        assert abs(res["1"][0].std().item() / res["0"][0].std().item() - 1.0) < 0.05
        return
    for a, b in zip(res["1"], res["0"]):
        torch.testing.assert_close(a, b, rtol=4e-2, atol=4e-2 * b.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1024, 4096, 8), (37, 11008, 8), (130, 256, 16), (5, 1032, 16)])
def test_lora_rank_kernels_match_fp32(shape, dtype):
    """lora_down / lora_up (one-pass rank-r products) against fp32 matmuls, both storage orders of the small operand,
    with and without the dropout mask and the accumulate flag; strided activations"""
    from bigdl_amd.ops import hip
    m, k, r = shape
    torch.manual_seed(30)
    x = torch.randn(m, k + 8, device="cuda").to(dtype)[:, :k]                   # row stride k + 8
    a = (torch.randn(r, k, device="cuda") * 0.05).to(dtype)                    # [r, K]   (an adapter's A)
    b = (torch.randn(k, r, device="cuda") * 0.05).to(dtype)                    # [K, r]   (an adapter's B, K = its out size)
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=4e-3, atol=4e-3)
    t1 = hip.lora_down(x, a, r)
    torch.testing.assert_close(t1.float(), x.float() @ a.float().t(), **tol)
    t2 = hip.lora_down(x, b, r, alpha=0.5)
    torch.testing.assert_close(t2.float(), 0.5 * (x.float() @ b.float()), **tol)
    t = (torch.randn(m, r, device="cuda") * 0.5).to(dtype)
    out0 = torch.randn(m, k + 8, device="cuda").to(dtype)[:, :k]
    out = out0.clone()
    hip.lora_up(t, a, out, alpha=2.0)                                           # out += 2 * t @ A
    torch.testing.assert_close(out.float(), out0.float() + 2.0 * (t.float() @ a.float()), **tol)
    out = out0.clone()
    hip.lora_up(t, b, out, alpha=1.0, accumulate=False)                         # out = t @ B^T
    torch.testing.assert_close(out.float(), t.float() @ b.float().t(), **tol)
    mask = torch.rand(m, k, device="cuda") > 0.3
    out = out0.clone()
    hip.lora_up(t, a, out, alpha=1.0 / 0.7, mask=mask)
    torch.testing.assert_close(out.float(), out0.float() + (t.float() @ a.float()) * mask.float() / 0.7, **tol)
    with pytest.raises(ValueError):
        hip.lora_down(x, a[:4], 4)


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_fused_lora_node_native_kernels_match_library_products(monkeypatch, p):
    """FusedLoraLowBit with the rank-r kernels against the same node on library GEMMs: same dropout kernel and seed, so
    the same mask; output, activation gradient and adapter gradients"""
    import bigdl_amd.transformers.qlora as Q_
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model
    pm = get_peft_model(_qmodel("nf4"), LoraConfig(r=8, lora_alpha=16, target_modules=["q_proj"], lora_dropout=p,
                                                   training_mode="qlora"))
    lin = pm.base_model.model.layers[0].self_attn.q_proj
    torch.nn.init.normal_(lin.lora_B["default"].weight, std=0.05)
    lin.train()
    x0 = torch.randn(3, 40, 256, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(3, 40, lin.out_features, device="cuda", dtype=torch.bfloat16)
    res = {}
    for native in (True, False):
        monkeypatch.setattr(Q_, "_LORA_KERNELS", native)
        x = x0.clone().requires_grad_(True)
        for prm in (lin.lora_A["default"].weight, lin.lora_B["default"].weight):
            prm.grad = None
        torch.manual_seed(123)
        y = lin(x)
        y.backward(g)
        res[native] = (y.detach().float(), x.grad.float(), lin.lora_A["default"].weight.grad.float().clone(),
                       lin.lora_B["default"].weight.grad.float().clone())
    for a, b in zip(res[True], res[False]):
        torch.testing.assert_close(a, b, rtol=4e-2, atol=4e-2 * b.abs().max().item())
