"""The decode kernels at the shapes and dispatch conditions of the headline model (Llama-2-7B: hidden 4096,
intermediate 11008, 32 heads x 128): every launch-time variant of the skinny GEMM (lean single-tile kernel, 16-wavefront
kernel for K >= 8192, gate/up pair, x fetched from global memory, attention-partial combine) and the split-KV attention
at a 1100-token cache, each against an fp32 PyTorch reference built from the host dequantizer (itself pinned to the
public ggml block formulas in tests/test_quantize_host.py)."""
import math

import pytest
import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.ops import host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from bigdl_amd.ops import hip as h
    h.lib()
    return h


def _quant_dev(w, qt):
    n, k = w.shape
    buf = host.quantize(w, qt)
    raw = buf.view(torch.uint8) if buf.dtype != torch.uint8 else buf
    ref = host.dequantize(raw, qt, w.shape)
    return host.relayout_to_device(raw, qt, n, k).cuda(), ref


def _close(out, ref, tol):
    # errors relative to the scale of the output row (fp16 inputs, fp32 accumulation)
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale, (err, scale)


# (N, K, M): lean kernel (one tile per workgroup, 8 wavefronts), 16-wavefront kernel (K >= 8192, N <= 4096), general
# kernel (3+ rows), and x read from global memory (16 rows of K = 11008 do not fit the LDS budget)
@pytest.mark.parametrize("n,k,m", [(12288, 4096, 1), (12288, 4096, 2), (4096, 4096, 1), (4096, 11008, 1), (4096, 11008, 2),
                                   (4096, 4096, 4), (4096, 11008, 4), (4096, 11008, 16), (32000, 4096, 1)])
@pytest.mark.parametrize("name", ["sym_int4", "asym_int4", "fp8_e5m2"])
def test_gemv_production_shapes(hip, name, n, k, m):
    if name != "sym_int4" and (n > 12288 or m == 16):
        pytest.skip("large cases once, on the headline format")
    torch.manual_seed(n + k + m)
    qt = Q.ggml_tensor_qtype[name]
    w = torch.randn(n, k) * 0.02
    dev, ref_w = _quant_dev(w, qt)
    x = torch.randn(m, k).half()
    gamma = (1 + 0.1 * torch.randn(k)).half()
    res = torch.randn(m, n).half()
    ref = x.float() @ ref_w.t()
    out = hip.gemv_lowbit(x.cuda(), dev, qt, n, k, out_dtype=torch.float32).cpu()
    _close(out, ref, 2e-3)
    # RMSNorm prologue + residual epilogue, fp16 output (what the decode step launches)
    xf = x.float()
    xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma.float()
    ref2 = xn @ ref_w.t() + res.float()
    out2 = hip.gemv_lowbit(x.cuda(), dev, qt, n, k, gamma=gamma.cuda(), eps=1e-5, residual=res.cuda()).cpu().float()
    _close(out2, ref2, 4e-3)


@pytest.mark.parametrize("m", [1, 2, 4])
def test_gate_up_production_shape(hip, m):
    torch.manual_seed(5 + m)
    qt = Q.SYM_INT4
    n, k = 11008, 4096
    dg, rg = _quant_dev(torch.randn(n, k) * 0.02, qt)
    du, ru = _quant_dev(torch.randn(n, k) * 0.02, qt)
    x = torch.randn(m, k).half()
    gamma = (1 + 0.1 * torch.randn(k)).half()
    xf = x.float()
    xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma.float()
    ref = torch.nn.functional.silu(xn @ rg.t()) * (xn @ ru.t())
    out = hip.gemv_lowbit(x.cuda(), dg, qt, n, k, w2=du, gamma=gamma.cuda(), eps=1e-5, act="silu",
                          out_dtype=torch.float32).cpu()
    _close(out, ref, 4e-3)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("batch", [1, 3])
def test_attention_1100_keys_32_heads_with_deferred_combine(hip, fp8, batch):
    """split-KV decode attention (fused RoPE + append) over 32 heads x 1100 cached keys, the partials combined inside the
    o_proj GEMV's staging: against fp32 attention + fp32 projection on the same (dequantized) cache contents."""
    torch.manual_seed(7)
    hq = hkv = 32
    d, past, lmax, splits = 128, 1100, 1280, 8
    hidden = hq * d
    qkv = (torch.randn(batch, 3 * hidden) * 0.5).half().cuda()
    kc = torch.randn(batch, hkv, lmax, d).half().cuda()
    vc = torch.randn(batch, hkv, d, lmax).half().cuda()
    if fp8:
        kq, vq = torch.empty_like(kc, dtype=torch.uint8), torch.empty_like(vc, dtype=torch.uint8)
        hip.to_e5m2(kc, kq)
        hip.to_e5m2(vc, vq)
        hip.from_e5m2(kq, kc)
        hip.from_e5m2(vq, vc)
        kcache, vcache = kq, vq
    else:
        kcache, vcache = kc.clone(), vc.clone()
    past_len = torch.full((batch,), past, dtype=torch.int32).cuda()
    max_pos = 2048
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    ang = torch.arange(max_pos).float()[:, None] * inv[None]
    table = torch.stack([ang.cos(), ang.sin()], -1).contiguous().cuda()          # [max_pos, d/2, 2]
    ws = torch.empty(hip.attn_decode_ws_floats(batch, hq, hkv, 1, d, splits) + 16, dtype=torch.float32).cuda()
    counters = torch.zeros(hip.attn_decode_counters(batch, hq, hkv), dtype=torch.int32).cuda()
    attn = torch.empty(batch, hidden, dtype=torch.float16).cuda()
    hip.attn_decode(qkv, kcache, vcache, past_len, attn, ws, counters, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=1,
                    batch=batch, scale=1 / math.sqrt(d), splits=splits, rope_table=table, defer_combine=True)
    qt = Q.SYM_INT4
    wo, ref_wo = _quant_dev(torch.randn(hidden, hidden) * 0.02, qt)
    h = torch.randn(batch, hidden).half().cuda()
    out = hip.gemv_lowbit(attn, wo, qt, hidden, hidden, residual=h,
                          att_partials=(ws, splits, d, 1, hkv, 1, 16)).cpu().float()

    # reference: rotate q and the new k (rotate-half pairs), append, softmax over past + 1 keys
    def rope(t):          # [B, H, D] at position `past`
        c, s = ang[past].cos(), ang[past].sin()
        t1, t2 = t[..., : d // 2], t[..., d // 2:]
        return torch.cat([t1 * c - t2 * s, t2 * c + t1 * s], -1)

    q3 = qkv.cpu().float().view(batch, 3, hq, d)
    q, kn, vn = rope(q3[:, 0]), rope(q3[:, 1]), q3[:, 2]
    if fp8:
        kn = kn.half().to(torch.float8_e5m2).float()
        vn = vn.half().to(torch.float8_e5m2).float()
    else:
        kn, vn = kn.half().float(), vn.half().float()
    k_all = torch.cat([kc.cpu().float()[:, :, :past], kn[:, :, None]], 2)                       # [B, H, L, D]
    v_all = torch.cat([vc.cpu().float()[:, :, :, :past], vn[:, :, :, None]], 3)                 # [B, H, D, L]
    p = torch.softmax(torch.einsum("bhd,bhld->bhl", q, k_all) / math.sqrt(d), -1)
    o = torch.einsum("bhl,bhdl->bhd", p, v_all).reshape(batch, hidden)
    ref = o.half().float() @ ref_wo.t() + h.cpu().float()
    _close(out, ref, 6e-3)
    # the appended key / value landed in the caches
    got_k = kcache[:, :, past].cpu()
    got_k = got_k.view(torch.float8_e5m2).float() if fp8 else got_k.float()
    _close(got_k, kn, 2e-2 if fp8 else 2e-3)


# ------------------------------------------------------------------------------------------------
# fp16 weights, row-major as stored (csrc/hip/gemv_f16.hip): decode / speculative-verification rows
@pytest.mark.parametrize("n,k,m", [(12288, 4096, 1), (4096, 4096, 5), (4096, 11008, 1), (4096, 11008, 9), (32000, 4096, 1),
                                    (4544, 4544, 3), (256, 64, 16),
                                    # row-stream form (1 - 2 rows): every request count per row, K % 512 != 0, x at the LDS cap
                                    (4096, 4096, 2), (4096, 11008, 2), (4096, 14336, 2), (8192, 8192, 2), (1024, 16384, 1),
                                    (4096, 11008, 4), (272, 4544, 2), (48, 64, 1), (1024, 16384, 2)])
def test_gemv_f16_against_fp32(n, k, m):
    from bigdl_amd.ops import hip
    g = torch.Generator(device="cuda").manual_seed(n + k + m)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).half()
    x = torch.randn(m, k, device="cuda", generator=g).half()
    gamma = (1 + 0.1 * torch.randn(k, device="cuda", generator=g)).half()
    res = torch.randn(m, n, device="cuda", generator=g).half()
    bias = (0.1 * torch.randn(n, device="cuda", generator=g)).half()
    xf = x.float()
    xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma.float()
    want = xn @ w.float().T + bias.float() + res.float()
    got = hip.gemv_f16(x, w, bias=bias, residual=res, gamma=gamma, eps=1e-5)
    err = (got.float() - want).abs().max().item()
    assert err < 2e-2 * max(1.0, want.abs().max().item()), err
    # no norm, no epilogue, fp32 out
    got = hip.gemv_f16(x, w, out_dtype=torch.float32)
    want = xf @ w.float().T
    assert (got - want).abs().max().item() < 5e-3 * max(1.0, want.abs().max().item())
    if m <= 2:
        # both forms of the kernel on the same operands
        hip.gemv_f16_tune(rows_form=0)
        try:
            other = hip.gemv_f16(x, w, out_dtype=torch.float32)
        finally:
            hip.gemv_f16_tune(rows_form=1)
        assert (got - other).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("m,n,k", [(1, 11008, 4096), (4, 11008, 4096), (9, 11008, 4096), (2, 28672, 8192), (2, 1024, 5120),
                                   (1, 512, 12288)])
def test_gemv_f16_gate_up(m, n, k):
    from bigdl_amd.ops import hip
    g = torch.Generator(device="cuda").manual_seed(m)
    wg = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).half()
    wu = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).half()
    x = torch.randn(m, k, device="cuda", generator=g).half()
    gamma = (1 + 0.1 * torch.randn(k, device="cuda", generator=g)).half()
    xf = x.float()
    xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * gamma.float()
    want = torch.nn.functional.silu(xn @ wg.float().T) * (xn @ wu.float().T)
    got = hip.gemv_f16(x, wg, w2=wu, gamma=gamma, eps=1e-6, act="silu")
    assert (got.float() - want).abs().max().item() < 2e-2 * max(1.0, want.abs().max().item())


def test_fp16_engine_decode_matches_its_prefill():
    """The fp16 engine (fused fp16 GEMVs per projection) decodes the same tokens as its own prompt pass predicts."""
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    cfg = DecoderConfig(vocab_size=4096, hidden_size=2048, intermediate_size=5632, num_layers=4, num_heads=16, num_kv_heads=8,
                        head_dim=128, max_position=2048)
    eng = build_random_decoder(cfg, Q.FP16, device="cuda", seed=11)
    assert not eng.layers[0].qkv.fused
    ids = torch.randint(3, cfg.vocab_size, (1, 40), generator=torch.Generator().manual_seed(0)).cuda()
    out = eng.generate(ids, max_new_tokens=10)
    # teacher-forced prompt pass over the generated sequence must predict the same continuation
    eng.ensure_cache(1, 64)
    eng.cache.reset()
    logits = eng.prefill(out[:, :-1], last_only=False)
    pred = logits[0, 39:].argmax(-1)
    assert (pred == out[0, 40:]).float().mean().item() >= 0.9


@pytest.mark.parametrize("mode", ["pair", "fold"])
@pytest.mark.parametrize("batch", [24, 64])
def test_batched_decode_fused_forms_match_separate_kernels(batch, mode, monkeypatch):
    """Decode at 17+ rows: gate / up as one launch with the activation in its reduce kernel (default), and the
    experimental form with the norm folded into the GEMM (BIGDL_FOLD_NORM=1), against separate kernels."""
    import torch
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    cfg = DecoderConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_layers=3, num_heads=8, num_kv_heads=4,
                        head_dim=64, max_position=256)
    eng = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=3)
    eng.use_graph = False
    ids = torch.randint(3, 512, (batch, 12)).cuda()
    outs = []
    for fused in (True, False):
        monkeypatch.delenv("BIGDL_FOLD_NORM", raising=False)
        monkeypatch.delenv("BIGDL_NO_PAIR_GEMM", raising=False)
        if fused and mode == "fold":
            monkeypatch.setenv("BIGDL_FOLD_NORM", "1")
        if not fused:
            monkeypatch.setenv("BIGDL_NO_PAIR_GEMM", "1")
        eng.ensure_cache(batch, 32)
        eng.cache.reset()
        logits = eng.prefill(ids)
        tok = logits[:, -1].argmax(-1).view(batch, 1)
        lg, _ = eng.decode(tok)
        lg2, _ = eng.decode(lg.view(batch, -1).argmax(-1).view(batch, 1))
        outs.append((lg.clone(), lg2.clone()))
        assert eng.fold_batched_norm is (fused and mode == "fold")
    for a, b in zip(outs[0], outs[1]):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2)
