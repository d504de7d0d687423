"""Fused training nodes (bigdl_amd/transformers/fused_train_ops.py): forward and input gradients against the plain
PyTorch fp32 expressions, and a LoRA training step of a tiny llama with and without them."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("h,offset", [(256, 0.0), (4096, 1.0), (1000, 0.0)])
def test_rms_norm_node(dtype, h, offset):
    from bigdl_amd.transformers.fused_train_ops import RMSNormFn
    torch.manual_seed(0)
    x = torch.randn(3, 7, h, device="cuda")
    w = (1.0 + 0.1 * torch.randn(h, device="cuda")) - offset
    g = torch.randn(3, 7, h, device="cuda")
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * (wr + offset)
    ref.backward(g)
    xt = x.to(dtype).requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    out = RMSNormFn.apply(xt, wt, 1e-5, offset)
    out.backward(g.to(dtype))
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out.float(), ref.detach(), **tol)
    torch.testing.assert_close(xt.grad.float(), xr.grad, **tol)
    torch.testing.assert_close(wt.grad.float(), wr.grad, rtol=3e-2, atol=3e-2 * (3 * 7) ** 0.5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_silu_mul_and_rope_nodes(dtype):
    from bigdl_amd.models.hf_patch import _rotate_half
    from bigdl_amd.transformers.fused_train_ops import RopeFn, SiluMulFn
    torch.manual_seed(1)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    a, u, g = (torch.randn(5, 11, 344, device="cuda") for _ in range(3))
    ar, ur = a.clone().requires_grad_(True), u.clone().requires_grad_(True)
    (F.silu(ar) * ur).backward(g)
    at, ut = a.to(dtype).requires_grad_(True), u.to(dtype).requires_grad_(True)
    out = SiluMulFn.apply(at, ut)
    out.backward(g.to(dtype))
    torch.testing.assert_close(out.float(), (F.silu(a) * u), **tol)
    torch.testing.assert_close(at.grad.float(), ar.grad, **tol)
    torch.testing.assert_close(ut.grad.float(), ur.grad, **tol)
    # rotary embedding on a strided [B, H, S, D] view of a fused projection
    b, s, hq, d = 2, 9, 4, 64
    qkv = torch.randn(b, s, 3 * hq, d, device="cuda")
    ang = torch.rand(b, s, d // 2, device="cuda") * 6
    cos, sin = torch.cat([ang.cos()] * 2, -1), torch.cat([ang.sin()] * 2, -1)
    gq = torch.randn(b, hq, s, d, device="cuda")
    qr = qkv.clone().requires_grad_(True)
    q_ref = qr.transpose(1, 2)[:, :hq]
    ref = q_ref * cos[:, None] + _rotate_half(q_ref) * sin[:, None]
    ref.backward(gq)
    qt = qkv.to(dtype).requires_grad_(True)
    out = RopeFn.apply(qt.transpose(1, 2)[:, :hq], cos.to(dtype), sin.to(dtype))
    out.backward(gq.to(dtype))
    torch.testing.assert_close(out.float(), ref.detach(), **tol)
    torch.testing.assert_close(qt.grad.float(), qr.grad, **tol)


def test_lora_step_with_fused_nodes_matches_plain_modules(monkeypatch):
    """One LoRA training step of a tiny llama: loss and adapter gradients with the nodes bound (get_peft_model does it)
    against the same model with BIGDL_FUSED_TRAIN_OPS=0."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.transformers.qlora import LoraConfig, get_peft_model

    def run(fused):
        monkeypatch.setenv("BIGDL_FUSED_TRAIN_OPS", "1" if fused else "0")
        torch.manual_seed(0)
        cfg = LlamaConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=4, max_position_embeddings=128)
        cfg._attn_implementation = "sdpa"
        base = LlamaForCausalLM(cfg)
        m = optimize_model(base, low_bit="nf4", optimize_llm=False).to("cuda")
        m.train()
        pm = get_peft_model(m, LoraConfig(r=4, lora_alpha=8, target_modules=["q_proj", "v_proj"], lora_dropout=0.0,
                                          bias="none", task_type="CAUSAL_LM", training_mode="qlora"))
        torch.manual_seed(1)
        for n_, p in pm.named_parameters():
            if "lora_B" in n_:
                p.data.normal_(0, 0.02)
        ids = torch.randint(3, 320, (2, 24), generator=torch.Generator().manual_seed(2)).cuda()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = pm(input_ids=ids, labels=ids).loss
        loss.backward()
        grads = {n_: p.grad.float().clone() for n_, p in pm.named_parameters() if p.grad is not None}
        return float(loss.detach()), grads, m
    loss_f, g_f, m_f = run(True)
    assert m_f.model.norm.forward.__func__.__name__ == "rms_norm_forward"
    assert m_f.model.layers[0].mlp.forward.__func__.__name__ == "mlp_forward"
    import sys
    assert getattr(sys.modules[type(m_f.model.layers[0].self_attn).__module__].apply_rotary_pos_emb, "_bigdl_fused", False)
    loss_p, g_p, _ = run(False)
    assert abs(loss_f - loss_p) < 2e-2 * abs(loss_p)
    assert g_f.keys() == g_p.keys() and len(g_f) >= 8
    for k in g_f:
        rel = (g_f[k] - g_p[k]).norm() / (g_p[k].norm() + 1e-12)
        assert rel < 0.08, (k, float(rel))
