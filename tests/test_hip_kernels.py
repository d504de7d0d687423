"""Numerics of the gfx950 kernels against plain PyTorch fp32 references of the same op."""
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
    dev = host.relayout_to_device(raw, qt, n, k).cuda()
    return dev, ref


T16 = ["sym_int4", "asym_int4", "woq_int4", "fp4", "nf4", "sym_int5", "asym_int5", "fp6", "sym_int8", "fp8_e4m3",
       "fp8_e5m2"]


@pytest.mark.parametrize("name", T16)
@pytest.mark.parametrize("shape", [(64, 256), (48, 384), (200, 1024)])
def test_dequant_t16(hip, name, shape):
    torch.manual_seed(0)
    qt = Q.ggml_tensor_qtype[name]
    n, k = shape
    w = torch.randn(n, k) * 0.05
    dev, ref = _quant_dev(w, qt)
    out = hip.dequant_t16(dev, qt, n, k, dtype=torch.float32).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-6)
    out16 = hip.dequant_t16(dev, qt, n, k, dtype=torch.float16).cpu().float()
    torch.testing.assert_close(out16, ref, rtol=2e-3, atol=1e-4)


@pytest.mark.parametrize("name", ["q4_k", "q5_k"])
@pytest.mark.parametrize("m", [1, 5, 40, 130])
def test_k_quants_run_on_the_asymmetric_tiled_kernels(hip, name, m):
    """q4_k / q5_k are tiled as per-32 asymmetric formats (scale = d * sc6, min = -dmin * m6 in fp16): GEMV, GEMM and
    dequantization on the device agree with the host dequantization of the original k-quant blocks."""
    torch.manual_seed(3)
    qt = Q.ggml_tensor_qtype[name]
    n, k = 80, 768
    w = torch.randn(n, k) * 0.05
    x = torch.randn(m, k).half()
    dev, ref_w = _quant_dev(w, qt)
    ref = x.float() @ ref_w.t()
    if m <= 16:
        out = hip.gemv_lowbit(x.cuda(), dev, qt, n, k, out_dtype=torch.float32).cpu()
    else:
        out = hip.gemm_lowbit(x.cuda(), dev, qt, n, k, out_dtype=torch.float32).cpu()
    torch.testing.assert_close(out, ref, rtol=4e-3, atol=4e-3)
    deq = hip.dequant_t16(dev, qt, n, k, dtype=torch.float32).cpu()
    torch.testing.assert_close(deq, ref_w, rtol=2e-3, atol=3e-4)


@pytest.mark.parametrize("name", T16)
@pytest.mark.parametrize("m", [1, 3, 16, 33])
def test_gemv_matches_dequant_reference(hip, name, m):
    torch.manual_seed(1)
    qt = Q.ggml_tensor_qtype[name]
    n, k = 96, 640
    w = torch.randn(n, k) * 0.05
    x = torch.randn(m, k)
    dev, ref_w = _quant_dev(w, qt)
    xh = x.half()
    ref = xh.float() @ ref_w.t()
    out = hip.gemv_lowbit(xh.cuda(), dev, qt, n, k, out_dtype=torch.float32).cpu()
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3)


def test_gemv_fused_norm_gate_up_bias_residual(hip):
    torch.manual_seed(2)
    qt = Q.SYM_INT4
    n, k, m = 176, 512, 2
    wg, wu = torch.randn(n, k) * 0.05, torch.randn(n, k) * 0.05
    dg, rg = _quant_dev(wg, qt)
    du, ru = _quant_dev(wu, qt)
    x = torch.randn(m, k).half()
    gamma = (1 + 0.1 * torch.randn(k)).half()
    eps = 1e-5
    xf = x.float()
    xn = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * gamma.float()
    ref = torch.nn.functional.silu(xn @ rg.t()) * (xn @ ru.t())
    out = hip.gemv_lowbit(x.cuda(), dg, qt, n, k, w2=du, gamma=gamma.cuda(), eps=eps, act="silu",
                          out_dtype=torch.float32).cpu()
    torch.testing.assert_close(out, ref, rtol=5e-3, atol=5e-3)
    # bias + residual, fp16 output
    bias = torch.randn(n).half()
    res = torch.randn(m, n).half()
    ref2 = xf @ rg.t() + bias.float() + res.float()
    out2 = hip.gemv_lowbit(x.cuda(), dg, qt, n, k, bias=bias.cuda(), residual=res.cuda(), out_dtype=torch.float16).cpu()
    torch.testing.assert_close(out2.float(), ref2, rtol=5e-3, atol=2e-2)


@pytest.mark.parametrize("name", ["q4_k", "q6_k", "q2_k", "q5_k", "nf3", "fp6_k", "sym_int4", "nf4"])
def test_dequant_rows_host_layout(hip, name):
    torch.manual_seed(3)
    qt = Q.ggml_tensor_qtype[name]
    w = torch.randn(50, 512) * 0.1
    buf = host.quantize(w, qt)
    raw = buf.view(torch.uint8) if buf.dtype != torch.uint8 else buf
    ref = host.dequantize(raw, qt, w.shape)
    out = hip.dequant_rows(raw.cuda(), qt, 50, 512, dtype=torch.float32).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    ids = torch.tensor([[3, 49, 0], [7, 7, 12]])
    out = hip.dequant_rows(raw.cuda(), qt, 50, 512, ids=ids.cuda(), dtype=torch.float32).cpu()
    torch.testing.assert_close(out, ref[ids], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_rms_norm_and_layer_norm(hip, dtype):
    torch.manual_seed(4)
    x = torch.randn(5, 7, 4096).to(dtype)
    w = (1 + 0.1 * torch.randn(4096)).to(dtype)
    b = (0.1 * torch.randn(4096)).to(dtype)
    xf = x.float()
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()
    out = hip.rms_norm(x.cuda(), w.cuda(), 1e-6).cpu().float()
    tol = dict(rtol=2e-2, atol=2e-2) if dtype != torch.float32 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out, ref, **tol)
    res = torch.randn_like(x)
    res_out = torch.empty_like(x).cuda()
    out = hip.rms_norm(x.cuda(), w.cuda(), 1e-6, residual=res.cuda(), res_out=res_out, offset=1.0).cpu().float()
    s = xf + res.float()
    ref = s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6) * (1 + w.float())
    torch.testing.assert_close(out, ref, **tol)
    torch.testing.assert_close(res_out.cpu().float(), s, **tol)
    ref = torch.nn.functional.layer_norm(xf, (4096,), w.float(), b.float(), 1e-5)
    out = hip.layer_norm(x.cuda(), w.cuda(), b.cuda(), 1e-5).cpu().float()
    torch.testing.assert_close(out, ref, **tol)


def test_act_mul_softmax_argmax_penalty(hip):
    torch.manual_seed(5)
    gu = torch.randn(6, 2 * 1024).half().cuda()
    gate, up = gu[:, :1024], gu[:, 1024:]
    ref = torch.nn.functional.silu(gate.float()) * up.float()
    out = torch.empty(6, 1024, dtype=torch.float16, device="cuda")
    hip.act_mul(gate, up, "silu", out=out)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)
    s = torch.randn(3, 4, 5, 333).cuda()
    ref = torch.softmax(s, -1)
    hip.softmax_inplace(s)
    torch.testing.assert_close(s, ref, rtol=1e-5, atol=1e-6)
    logits = torch.randn(4, 32000).half().cuda()
    assert torch.equal(hip.argmax(logits), logits.float().argmax(-1))
    scores = torch.randn(2, 1000).cuda()
    ids = torch.randint(0, 1000, (2, 17)).cuda()
    ref = scores.clone()
    g = torch.gather(ref, 1, ids)
    ref.scatter_(1, ids, torch.where(g < 0, g * 1.3, g / 1.3))
    hip.repetition_penalty_inplace(scores, ids, 1.3)
    torch.testing.assert_close(scores, ref)


@pytest.mark.parametrize("interleaved", [False, True])
def test_rope_inplace_strided(hip, interleaved):
    torch.manual_seed(6)
    b, h, s, d, rot = 2, 4, 9, 128, 64
    qkv = torch.randn(b, s, 3 * h * d).half().cuda()
    q = qkv[..., : h * d].view(b, s, h, d).transpose(1, 2)  # strided view, like the fused-qkv split
    q_ref = q.float().clone()
    pos = torch.arange(5, 5 + s)[None].expand(b, s).cuda()
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot)).cuda()
    ang = pos[:, None, :, None].float() * inv_freq
    cos, sin = ang.cos(), ang.sin()
    xr = q_ref[..., :rot]
    if interleaved:
        x0, x1 = xr[..., 0::2], xr[..., 1::2]
        y = torch.stack([x0 * cos - x1 * sin, x1 * cos + x0 * sin], -1).flatten(-2)
    else:
        x0, x1 = xr[..., : rot // 2], xr[..., rot // 2:]
        y = torch.cat([x0 * cos - x1 * sin, x1 * cos + x0 * sin], -1)
    ref = torch.cat([y, q_ref[..., rot:]], -1)
    hip.rope_inplace(q, position_ids=pos, inv_freq=inv_freq, interleaved=interleaved, rot_dim=rot)
    torch.testing.assert_close(q.float(), ref, rtol=2e-3, atol=2e-3)


def test_e5m2_roundtrip(hip):
    x = (torch.randn(3, 5, 128) * 4).half().cuda()
    q = torch.empty(3, 5, 128, dtype=torch.uint8, device="cuda")
    hip.to_e5m2(x, q)
    ref = x.to(torch.float8_e5m2)
    assert torch.equal(q.view(torch.float8_e5m2).float(), ref.float())
    back = torch.empty_like(x)
    hip.from_e5m2(q, back)
    assert torch.equal(back.float(), ref.float())


def _ref_attention(q, k, v, scale, softcap=0.0, window=0, past=0):
    # q [B,Hq,S,D], k/v [B,Hkv,L,D]; causal over the last S positions
    b, hq, s, d = q.shape
    hkv, L = k.shape[1], k.shape[2]
    k = k.repeat_interleave(hq // hkv, 1)
    v = v.repeat_interleave(hq // hkv, 1)
    sc = (q @ k.transpose(-1, -2)) * scale
    if softcap:
        sc = softcap * torch.tanh(sc / softcap)
    qpos = past + torch.arange(s)[:, None]
    kpos = torch.arange(L)[None]
    m = kpos > qpos
    if window:
        m |= kpos < qpos - window + 1
    sc = sc.masked_fill(m, float("-inf"))
    return torch.softmax(sc, -1) @ v


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("cfg", [(1, 32, 32, 128, 1, 1000), (2, 32, 8, 128, 1, 300), (1, 8, 2, 64, 5, 77),
                                 (1, 4, 4, 256, 3, 513), (2, 6, 2, 96, 2, 130),
                                 # multi-query groups above 32 query heads per kv head (starcoder 48, falcon-7b 71)
                                 (1, 48, 1, 128, 1, 300), (2, 71, 1, 64, 2, 130),
                                 # enough (batch, head) pairs for one KV split: 64 keys per wavefront on e5m2 caches
                                 (4, 8, 8, 128, 1, 1100), (3, 4, 2, 256, 2, 700)])
def test_rope_append_and_attn_decode(hip, fp8, cfg):
    torch.manual_seed(7)
    b, hq, hkv, d, q_len, past = cfg
    lmax = 1280
    ld = (hq + 2 * hkv) * d
    cache_dtype = torch.uint8 if fp8 else torch.float16
    kc = torch.zeros(b, hkv, lmax, d, dtype=cache_dtype, device="cuda")
    vc = torch.zeros(b, hkv, d, lmax, dtype=cache_dtype, device="cuda")
    inv_freq = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))

    def rope(x, pos):  # x [B,H,S,D]
        ang = pos[None, None, :, None].float() * inv_freq
        x0, x1 = x[..., : d // 2], x[..., d // 2:]
        return torch.cat([x0 * ang.cos() - x1 * ang.sin(), x1 * ang.cos() + x0 * ang.sin()], -1)

    def q8(t):
        return t.half().to(torch.float8_e5m2).float() if fp8 else t.half().float()

    # fill the past through the same kernel in chunks (exercises append at several positions)
    k_all = torch.zeros(b, hkv, past + q_len, d)
    v_all = torch.zeros(b, hkv, past + q_len, d)
    pos0 = 0
    past_len = torch.zeros(b, dtype=torch.int32, device="cuda")
    chunks = [c for c in (past // 2, past - past // 2, q_len) if c > 0]
    for ci, c in enumerate(chunks):
        qkv = (torch.randn(b * c, ld) * 0.5).half()
        x = qkv.float().view(b, c, hq + 2 * hkv, d).transpose(1, 2)
        pos = torch.arange(pos0, pos0 + c)
        k_all[:, :, pos0:pos0 + c] = q8(rope(x[:, hq:hq + hkv], pos))
        v_all[:, :, pos0:pos0 + c] = q8(x[:, hq + hkv:])
        qkv_d = qkv.cuda()
        past_len.fill_(pos0)
        hip.rope_append(qkv_d, kc, vc, past_len, inv_freq.cuda(), n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=c,
                        batch=b)
        q_rot = rope(x[:, :hq], pos)
        got_q = qkv_d.cpu().float().view(b, c, hq + 2 * hkv, d).transpose(1, 2)[:, :hq]
        torch.testing.assert_close(got_q, q_rot, rtol=3e-3, atol=3e-3)
        last = (qkv_d, q_rot, c, pos0)
        pos0 += c
    # cache contents
    if fp8:
        kc_f = kc.view(torch.float8_e5m2).float().cpu()
        vc_f = vc.view(torch.float8_e5m2).float().cpu()
    else:
        kc_f, vc_f = kc.float().cpu(), vc.float().cpu()
    tol = dict(rtol=0.13, atol=2e-2) if fp8 else dict(rtol=3e-3, atol=3e-3)
    if fp8:
        # device sincosf vs torch can flip an e5m2 rounding (one ulp = 25 %) on a handful of entries
        bad = ((kc_f[:, :, :pos0] - k_all).abs() > 0.13 * k_all.abs() + 2e-2).float().mean()
        assert bad < 1e-5, bad
    else:
        torch.testing.assert_close(kc_f[:, :, :pos0], k_all, **tol)
    torch.testing.assert_close(vc_f[:, :, :, :pos0].transpose(-1, -2), v_all, **tol)

    qkv_d, q_rot, c, p0 = last
    scale = 1.0 / math.sqrt(d)
    for splits, softcap, window in ((1, 0.0, 0), (4, 0.0, 0), (3, 30.0, 0), (2, 0.0, 64)):
        out = torch.empty(b * c, hq * d, dtype=torch.float16, device="cuda")
        ws = torch.empty(hip.attn_decode_ws_floats(b, hq, hkv, c, d, splits), dtype=torch.float32, device="cuda")
        counters = torch.zeros(hip.attn_decode_counters(b, hq, hkv), dtype=torch.int32, device="cuda")
        past_len.fill_(p0)
        hip.attn_decode(qkv_d, kc, vc, past_len, out, ws, counters, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=c,
                        batch=b, scale=scale, splits=splits, softcap=softcap, window=window)
        # reference on exactly what the cache holds
        ref = _ref_attention(q_rot.half().float(), kc_f[:, :, :pos0], vc_f[:, :, :, :pos0].transpose(-1, -2), scale,
                             softcap, window, past=p0)
        got = out.cpu().float().view(b, c, hq, d).transpose(1, 2)
        torch.testing.assert_close(got, ref, rtol=1e-2, atol=1e-2)
        assert int(counters.abs().sum()) == 0


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("splits", [3, 1])
@pytest.mark.parametrize("cfg", [(1, 32, 32, 128, 1, 700), (2, 8, 2, 128, 3, 255), (1, 4, 4, 64, 6, 254), (1, 16, 2, 128, 9, 40)])
def test_attn_decode_fused_rope_append_equals_separate(hip, fp8, interleaved, cfg, splits):
    """One launch (rope in registers + append inside the attention kernel) == rope_append kernel + attention."""
    torch.manual_seed(8)
    b, hq, hkv, d, q_len, past = cfg
    lmax = 1024
    ld = (hq + 2 * hkv) * d
    dt = torch.uint8 if fp8 else torch.float16
    inv_freq = (1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))).cuda()
    kc = (torch.randn(b, hkv, lmax, d) * 0.5).half().cuda()
    vc = (torch.randn(b, hkv, d, lmax) * 0.5).half().cuda()
    if fp8:
        kq, vq = torch.empty_like(kc, dtype=torch.uint8), torch.empty_like(vc, dtype=torch.uint8)
        hip.to_e5m2(kc, kq)
        hip.to_e5m2(vc, vq)
        kc, vc = kq, vq
    kc2, vc2 = kc.clone(), vc.clone()
    qkv = (torch.randn(b * q_len, ld) * 0.5).half().cuda()
    qkv2 = qkv.clone()
    past_len = torch.full((b,), past, dtype=torch.int32, device="cuda")
    scale = 1.0 / math.sqrt(d)
    ws = torch.empty(hip.attn_decode_ws_floats(b, hq, hkv, q_len, d, splits) + 16, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(b * hkv, dtype=torch.int32, device="cuda")
    out1 = torch.empty(b * q_len, hq * d, dtype=torch.float16, device="cuda")
    hip.rope_append(qkv, kc, vc, past_len, inv_freq, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=q_len, batch=b,
                    interleaved=interleaved)
    hip.attn_decode(qkv, kc, vc, past_len, out1, ws, cnt, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=q_len, batch=b,
                    scale=scale, splits=splits)
    pos = torch.arange(lmax, device="cuda").float()
    ang = pos[:, None] * inv_freq[None]
    table = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
    out2 = torch.empty_like(out1)
    hip.attn_decode(qkv2, kc2, vc2, past_len, out2, ws, cnt, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=q_len,
                    batch=b, scale=scale, splits=splits, rope_table=table, interleaved=interleaved)
    torch.testing.assert_close(out2.float(), out1.float(), rtol=5e-3, atol=5e-3)
    assert torch.equal(vc2, vc)
    if fp8:
        diff = (kc2.view(torch.float8_e5m2).float() - kc.view(torch.float8_e5m2).float()).abs()
        # table cos/sin vs sincosf can flip an e5m2 rounding: allow one ulp (25 %) on a few entries
        assert (diff > 0).float().mean() < 0.02
    else:
        torch.testing.assert_close(kc2.float(), kc.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("cfg", [(1, 32, 32, 128, 1, 1000, 5), (2, 8, 2, 128, 2, 300, 3), (1, 4, 4, 64, 3, 513, 2)])
def test_gemv_combines_attention_partials(hip, cfg):
    """o_proj GEMV fed with un-combined split-KV partials == attention with in-kernel combine + plain GEMV."""
    torch.manual_seed(9)
    b, hq, hkv, d, q_len, past, splits = cfg
    lmax = 1280
    ld = (hq + 2 * hkv) * d
    kc = (torch.randn(b, hkv, lmax, d) * 0.5).half().cuda()
    vc = (torch.randn(b, hkv, d, lmax) * 0.5).half().cuda()
    qkv = (torch.randn(b * q_len, ld) * 0.5).half().cuda()
    past_len = torch.full((b,), past, dtype=torch.int32, device="cuda")
    ws = torch.zeros(hip.attn_decode_ws_floats(b, hq, hkv, q_len, d, splits) + 16, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(b * hkv, dtype=torch.int32, device="cuda")
    att = torch.empty(b * q_len, hq * d, dtype=torch.float16, device="cuda")
    kw = dict(n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=q_len, batch=b, scale=1.0 / math.sqrt(d), splits=splits)
    hip.attn_decode(qkv, kc, vc, past_len, att, ws, cnt, **kw)
    n, k = 96, hq * d
    w = torch.randn(n, k) * 0.05
    dev, _ = _quant_dev(w, Q.SYM_INT4)
    ref = hip.gemv_lowbit(att, dev, Q.SYM_INT4, n, k, out_dtype=torch.float32)
    ws.zero_()
    dummy = torch.zeros_like(att)
    hip.attn_decode(qkv, kc, vc, past_len, dummy, ws, cnt, defer_combine=True, **kw)
    g = hq // hkv
    got = hip.gemv_lowbit(dummy, dev, Q.SYM_INT4, n, k, out_dtype=torch.float32,
                          att_partials=(ws, splits, d, g, hkv, q_len, (g * q_len + 15) // 16 * 16))
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)


def test_e5m2_into_cache_slices(hip):
    """Writes into / reads from a [B, H, n, D] slice of a larger cache buffer (no single row stride)."""
    torch.manual_seed(23)
    b, h, cap, d, n = 2, 3, 40, 64, 7
    x = torch.randn(b, h, n, d).half().cuda()
    cache = torch.zeros(b, h, cap, d, dtype=torch.uint8, device="cuda")
    hip.to_e5m2(x, cache[:, :, 5:5 + n])
    ref = x.to(torch.float8_e5m2).view(torch.uint8)
    assert torch.equal(cache[:, :, 5:5 + n], ref) and int(cache[:, :, :5].sum()) == 0
    out = torch.zeros(b, h, cap, d, dtype=torch.float16, device="cuda")
    hip.from_e5m2(cache[:, :, 5:5 + n], out[:, :, 5:5 + n])
    torch.testing.assert_close(out[:, :, 5:5 + n], ref.view(torch.float8_e5m2).half())
    xt = x.transpose(-1, -2).contiguous()                      # V^T layout: [B, H, D, n] slice of [B, H, D, cap]
    vc = torch.zeros(b, h, d, cap, dtype=torch.uint8, device="cuda")
    hip.to_e5m2(xt, vc[:, :, :, 5:5 + n])
    assert torch.equal(vc[:, :, :, 5:5 + n], xt.to(torch.float8_e5m2).view(torch.uint8))


# ---------------------------------------------------------------------------------------------
# mixture of experts
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [(5, 8, 2, True), (3, 60, 4, False), (1, 128, 8, True), (9, 4, 1, True)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_moe_route_matches_softmax_topk(hip, cfg, dtype):
    t, e, k, norm = cfg
    torch.manual_seed(20)
    logits = (torch.randn(t, e) * 2).to(dtype).cuda()
    ids, w = hip.moe_route(logits, k, normalize=norm)
    p = torch.softmax(logits.float(), -1)
    rv, ri = torch.topk(p, k, dim=-1)
    if norm:
        rv = rv / rv.sum(-1, keepdim=True)
    assert torch.equal(ids.long(), ri)
    torch.testing.assert_close(w, rv, rtol=1e-4, atol=1e-6)
    # expert-parallel view: ids outside [lo, hi) are -1, inside are rebased; weights unchanged
    lo, hi = e // 2, e
    ids2, w2 = hip.moe_route(logits, k, normalize=norm, expert_range=(lo, hi))
    exp = torch.where((ri >= lo) & (ri < hi), ri - lo, torch.full_like(ri, -1))
    assert torch.equal(ids2.long(), exp)
    torch.testing.assert_close(w2, rv, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("cfg", [(1, 4096, 8, 2, True, 0.0), (5, 2048, 60, 4, False, 0.0), (3, 8192, 32, 6, True, 1.0),
                                 (17, 256, 130, 8, True, 0.0)])
def test_moe_norm_route_matches_separate_ops(hip, cfg):
    """one launch (RMSNorm + router product + top-k) against a plain fp32 PyTorch statement of the three steps, and
    on x against the stand-alone norm kernel"""
    t, h, e, k, norm, offset = cfg
    torch.manual_seed(22)
    hid = torch.randn(t + 1, h + 8).half().cuda()[1:, :h]            # strided rows
    gamma = (1 + 0.1 * torch.randn(h)).half().cuda()
    router = (torch.randn(e, h) * 0.05).half().cuda()
    x, ids, w = hip.moe_norm_route(hid, gamma, 1e-5, router, k, normalize=norm, offset=offset)
    torch.testing.assert_close(x, hip.rms_norm(hid, gamma, 1e-5, offset=offset), rtol=1e-3, atol=1e-3)   # (summation order)
    hf = hid.float()
    xr = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5) * (gamma.float() + offset)).half()
    torch.testing.assert_close(x, xr, rtol=2e-3, atol=2e-3)
    logits = (x.float() @ router.float().t()).half().float()        # the dense product writes fp16 logits
    p = torch.softmax(logits, -1)
    rv, ri = torch.topk(p, k, dim=-1)
    if norm:
        rv = rv / rv.sum(-1, keepdim=True)
    # a different summation order can move a logit by one fp16 step: compare the picked probability mass, and the ids
    # wherever the reference's k-th and (k+1)-th candidates are not within such a step of each other
    pk = p.gather(1, ids.long())
    if norm:
        pk = pk / pk.sum(-1, keepdim=True)
    torch.testing.assert_close(w, pk, rtol=2e-3, atol=1e-5)
    srt = p.sort(-1, descending=True).values
    clear = (srt[:, k - 1] - srt[:, k]) > 4e-3 * srt[:, k - 1]      # the set of picked experts is unambiguous
    assert clear.any()
    assert torch.equal(ids.long().sort(-1).values[clear], ri.sort(-1).values[clear])
    lo, hi = e // 2, e
    _, ids2, w2 = hip.moe_norm_route(hid, gamma, 1e-5, router, k, normalize=norm, offset=offset, expert_range=(lo, hi))
    il = ids.long()
    assert torch.equal(ids2.long(), torch.where((il >= lo) & (il < hi), il - lo, torch.full_like(il, -1)))
    assert torch.equal(w2, w)


@pytest.mark.parametrize("cfg", [(4, 512, 64, 6, 8, 3, False, False, False, False),      # DeepSeek-V2 style: softmax, max
                                 (3, 1024, 160, 6, 8, 3, False, False, False, False),
                                 (5, 1024, 256, 8, 8, 4, True, True, True, True),         # V3 style: sigmoid + bias, top-2 sums
                                 (2, 512, 64, 6, 1, 1, True, True, False, True),          # bias only, one group
                                 (2, 512, 96, 4, 4, 2, True, False, True, False)])
def test_moe_norm_route_group_limited(hip, cfg):
    """the group-limited / biased selection inside the fused launch against the same steps as tensor ops"""
    t, h, e, k, n_group, topk_group, sigmoid, with_bias, top2, norm = cfg
    torch.manual_seed(24)
    hid = torch.randn(t, h).half().cuda()
    gamma = (1 + 0.1 * torch.randn(h)).half().cuda()
    router = (torch.randn(e, h) * 0.08).half().cuda()
    bias = (torch.randn(e) * 0.2).float().cuda() if with_bias else None
    x, ids, w = hip.moe_norm_route(hid, gamma, 1e-6, router, k, normalize=norm, sigmoid_scores=sigmoid, route_bias=bias,
                                   n_group=n_group, topk_group=topk_group, group_top2=top2)
    logits = (x.float() @ router.float().t()).half().float()
    p = torch.sigmoid(logits) if sigmoid else torch.softmax(logits, -1)
    choice = p if bias is None else p + bias
    if n_group > 1:
        grouped = choice.view(t, n_group, -1)
        gsc = grouped.topk(2, dim=-1).values.sum(-1) if top2 else grouped.amax(-1)
        keep = torch.zeros_like(gsc).scatter_(1, gsc.topk(topk_group, dim=-1).indices, 1.0)
        choice = choice.masked_fill(keep.repeat_interleave(e // n_group, 1) == 0, 0.0)
    rv, ri = torch.topk(choice, k, dim=-1)
    # rows whose selection does not hang on a rounding-sized gap (k-th against (k+1)-th candidate, and the group cut)
    srt = choice.sort(-1, descending=True).values
    clear = (srt[:, k - 1] - srt[:, k]) > 2e-3 * srt[:, k - 1].abs().clamp_min(1e-3)
    if n_group > 1:
        gsrt = gsc.sort(-1, descending=True).values
        clear &= (gsrt[:, topk_group - 1] - gsrt[:, topk_group]) > 2e-3 * gsrt[:, topk_group - 1].abs().clamp_min(1e-3)
    assert clear.any()
    assert torch.equal(ids.long().sort(-1).values[clear], ri.sort(-1).values[clear])
    wr = p.gather(1, ids.long())
    if norm:
        wr = wr / (wr.sum(-1, keepdim=True) + 1e-20)
    torch.testing.assert_close(w, wr, rtol=2e-3, atol=1e-6)
    # picks are ordered best first, like torch.topk
    cg = choice.gather(1, ids.long())
    assert bool((cg[clear][:, :-1] >= cg[clear][:, 1:] - 1e-6).all())
    # the same selection from given logits (the form the prompt pass and large routers use): same inputs as the
    # reference, so ids and order match exactly on the clear rows
    for lg in (logits.half(), logits):
        ids2, w2 = hip.moe_route(lg, k, normalize=norm, sigmoid_scores=sigmoid, route_bias=bias, n_group=n_group,
                                 topk_group=topk_group, group_top2=top2)
        assert torch.equal(ids2.long()[clear], ri[clear])
        wr2 = p.gather(1, ids2.long())
        if norm:
            wr2 = wr2 / (wr2.sum(-1, keepdim=True) + 1e-20)
        torch.testing.assert_close(w2, wr2, rtol=1e-4, atol=1e-6)
    lo, hi = e // 2, e
    ids3, _ = hip.moe_route(logits, k, normalize=norm, sigmoid_scores=sigmoid, route_bias=bias, n_group=n_group,
                            topk_group=topk_group, group_top2=top2, expert_range=(lo, hi))
    il = ids2.long()
    assert torch.equal(ids3.long(), torch.where((il >= lo) & (il < hi), il - lo, torch.full_like(il, -1)))


@pytest.mark.parametrize("name", ["sym_int4", "asym_int4", "nf4", "sym_int8", "fp6", "fp8_e5m2"])
@pytest.mark.parametrize("shape", [(6, 256, 96, 3, 2), (8, 384, 160, 1, 3), (10, 128, 64, 4, 8), (8, 14336, 64, 1, 2)])
def test_routed_expert_sum(hip, name, shape):
    """a token's routed down products summed inside one launch (with the residual) against fp32 products over the
    dequantized expert matrices"""
    torch.manual_seed(23)
    qt = Q.ggml_tensor_qtype[name]
    e, i, h, t, k = shape
    down = torch.randn(e * h, i) * 0.05
    dd, dr = _quant_dev(down, qt)
    a = (torch.randn(t * k, i) * 0.5).half()
    ids = torch.stack([torch.randperm(e)[:k] for _ in range(t)]).int()
    if t > 1:
        ids[1, 0] = -1                      # expert of another rank: its row of x is zero by contract
        a[k] = 0
    res = torch.randn(t, h).half()
    ref = res.float().clone()
    for s_ in range(t * k):
        ex = int(ids.view(-1)[s_])
        if ex >= 0:
            ref[s_ // k] += dr[ex * h:(ex + 1) * h] @ a[s_].float()
    out = hip.gemv_lowbit_moe_sum(a.cuda(), dd, qt, e, h, i, ids.view(-1).cuda(), k, residual=res.cuda())
    torch.testing.assert_close(out.float().cpu(), ref, rtol=5e-3, atol=2e-2 if i > 1000 else 5e-3)
    inplace = res.cuda()
    assert hip.gemv_lowbit_moe_sum(a.cuda(), dd, qt, e, h, i, ids.view(-1).cuda(), k, residual=inplace, out=inplace) is inplace
    assert torch.equal(inplace, out)
    plain = hip.gemv_lowbit_moe_sum(a.cuda(), dd, qt, e, h, i, ids.view(-1).cuda(), k)
    torch.testing.assert_close(plain.float().cpu(), ref - res.float(), rtol=5e-3, atol=2e-2 if i > 1000 else 5e-3)


@pytest.mark.parametrize("name", ["sym_int4", "asym_int4", "sym_int8", "fp6"])
def test_routed_expert_gemv(hip, name):
    torch.manual_seed(21)
    qt = Q.ggml_tensor_qtype[name]
    e, i, h, t, k = 6, 96, 256, 3, 2
    gate, up, down = torch.randn(e * i, h) * 0.05, torch.randn(e * i, h) * 0.05, torch.randn(e * h, i) * 0.05
    gd, gr = _quant_dev(gate, qt)
    ud, ur = _quant_dev(up, qt)
    dd, dr = _quant_dev(down, qt)
    x = torch.randn(t, h).half()
    ids = torch.tensor([[0, 5], [3, 3], [2, -1]], dtype=torch.int32)
    wts = torch.rand(t, k)
    a = hip.gemv_lowbit_moe(x.cuda(), gd, qt, e, i, h, ids.view(-1).cuda(), w2=ud, slot_xdiv=k, act="silu")
    y = hip.gemv_lowbit_moe(a, dd, qt, e, h, i, ids.view(-1).cuda(), slot_scale=wts.view(-1).cuda())
    ref_a = torch.zeros(t * k, i)
    ref_y = torch.zeros(t * k, h)
    for s in range(t * k):
        ex = int(ids.view(-1)[s])
        if ex < 0:
            continue
        xs = x[s // k].float()
        g = gr[ex * i:(ex + 1) * i] @ xs
        u = ur[ex * i:(ex + 1) * i] @ xs
        ref_a[s] = torch.nn.functional.silu(g) * u
        ref_y[s] = (dr[ex * h:(ex + 1) * h] @ a[s].float().cpu()) * wts.view(-1)[s]
    torch.testing.assert_close(a.float().cpu(), ref_a, rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(y.float().cpu(), ref_y, rtol=5e-3, atol=5e-3)
    # fused RMSNorm prologue per slot
    gamma = (1 + 0.1 * torch.randn(h)).half()
    a2 = hip.gemv_lowbit_moe(x.cuda(), gd, qt, e, i, h, ids.view(-1).cuda(), w2=ud, slot_xdiv=k, act="silu",
                             gamma=gamma.cuda(), eps=1e-5)
    xn = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5) * gamma.float())
    s = 1
    ex = int(ids.view(-1)[s])
    ref = torch.nn.functional.silu(gr[ex * i:(ex + 1) * i] @ xn[0]) * (ur[ex * i:(ex + 1) * i] @ xn[0])
    torch.testing.assert_close(a2[s].float().cpu(), ref, rtol=2e-2, atol=2e-2)
    # row-range dequant of one expert
    de = hip.dequant_t16(gd, qt, i, h, dtype=torch.float32, total_rows=e * i, row0=4 * i).cpu()
    torch.testing.assert_close(de, gr[4 * i:5 * i], rtol=1e-3, atol=1e-6)


def test_moe_combine(hip):
    torch.manual_seed(22)
    t, k, h = 5, 3, 264
    y = torch.randn(t * k, h).half().cuda()
    sh = torch.randn(t, h).half().cuda()
    gl = torch.randn(t, 1).half().cuda()
    res = torch.randn(t, h).half().cuda()
    out = torch.empty(t, h, dtype=torch.float16, device="cuda")
    hip.moe_combine(y, k, out, shared=sh, shared_gate=gl, residual=res)
    ref = y.float().view(t, k, h).sum(1) + torch.sigmoid(gl.float()) * sh.float() + res.float()
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)
    hip.moe_combine(y, k, res, residual=res)            # in place on the residual stream
    torch.testing.assert_close(res.float(), (ref - torch.sigmoid(gl.float()) * sh.float()), rtol=2e-3, atol=4e-3)


# ---------------------------------------------------------------------------------------------
# prefill attention
# ---------------------------------------------------------------------------------------------
def _sdpa_ref(q, k, v, scale, causal, mask=None, softcap=0.0, window=0):
    b, hq, sq, d = q.shape
    hkv, lk = k.shape[1], k.shape[2]
    kk = k.float().repeat_interleave(hq // hkv, 1)
    vv = v.float().repeat_interleave(hq // hkv, 1)
    s = torch.matmul(q.float(), kk.transpose(-1, -2)) * scale
    if softcap:
        s = softcap * torch.tanh(s / softcap)
    if mask is not None:
        s = s + mask.float()
    qi = torch.arange(sq, device=q.device)[:, None] + (lk - sq)
    ki = torch.arange(lk, device=q.device)[None]
    ok = torch.ones(sq, lk, dtype=torch.bool, device=q.device)
    if causal:
        ok &= ki <= qi
    if window:
        ok &= ki > qi - window
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    return torch.matmul(p, vv).transpose(1, 2).reshape(b, sq, hq * d)


@pytest.mark.parametrize("cfg", [(1, 8, 8, 128, 200, 200, True), (2, 8, 2, 64, 77, 77, True), (1, 4, 4, 128, 64, 300, True),
                                 (1, 4, 2, 128, 130, 130, False), (2, 4, 1, 64, 33, 90, False),
                                 (1, 6, 6, 96, 150, 150, True), (2, 4, 2, 96, 70, 133, True), (1, 4, 4, 96, 65, 65, False),
                                 (1, 4, 2, 256, 140, 140, True), (2, 2, 1, 256, 33, 100, False)])
def test_attn_prefill_matches_reference(hip, cfg):
    b, hq, hkv, d, sq, lk, causal = cfg
    torch.manual_seed(30)
    # strided views into a fused qkv buffer, as the engine passes them
    qkv = torch.randn(b, lk, (hq + 2 * hkv) * d, device="cuda").half()
    q = qkv[:, lk - sq:, : hq * d].view(b, sq, hq, d).transpose(1, 2)
    k = qkv[..., hq * d:(hq + hkv) * d].view(b, lk, hkv, d).transpose(1, 2)
    v = qkv[..., (hq + hkv) * d:].view(b, lk, hkv, d).transpose(1, 2)
    scale = d ** -0.5
    out = hip.attn_prefill(q, k, v, scale=scale, causal=causal)
    ref = _sdpa_ref(q, k, v, scale, causal)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-3)
    # V^T input (the engine's cache layout): a [B, Hkv, D, cap] buffer of which the first lk keys are valid
    cap = (lk + 255) // 256 * 256
    vt = torch.zeros(b, hkv, d, cap, device="cuda", dtype=torch.float16)
    vt[..., :lk] = v.transpose(-1, -2)
    out2 = hip.attn_prefill(q, k, vt[..., :lk], scale=scale, causal=causal, v_transposed=True)
    torch.testing.assert_close(out2.float(), ref, rtol=2e-2, atol=2e-3)


def test_attn_prefill_mask_softcap_window(hip):
    torch.manual_seed(31)
    b, hq, hkv, d, s = 2, 4, 2, 128, 150
    q, k, v = (torch.randn(b, h, s, d, device="cuda").half() for h in (hq, hkv, hkv))
    scale = d ** -0.5
    out = hip.attn_prefill(q, k, v, scale=scale, causal=True, softcap=30.0, window=40)
    torch.testing.assert_close(out.float(), _sdpa_ref(q, k, v, scale, True, softcap=30.0, window=40), rtol=2e-2, atol=2e-3)
    # left-padded batch: additive mask hides the first keys of row 1; fully masked query rows give zeros
    mask = torch.zeros(b, 1, s, s, device="cuda", dtype=torch.float16)
    mask[1, :, :, :20] = float("-inf")
    out = hip.attn_prefill(q, k, v, scale=scale, causal=True, mask=mask)
    ref = _sdpa_ref(q, k, v, scale, True, mask=mask)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-3)
    assert float(out[1, :20].abs().max()) == 0.0


def _alibi_slopes(n):
    base = 2.0 ** (-8.0 / n)
    return torch.tensor([base ** (i + 1) for i in range(n)], dtype=torch.float32)


@pytest.mark.parametrize("cfg", [(1, 8, 8, 128, 200, 200), (2, 8, 2, 64, 77, 77), (1, 4, 4, 128, 64, 300)])
def test_attn_prefill_alibi_and_per_head_mask(hip, cfg):
    """ALiBi slopes inside the kernel == the same bias passed as a per-head additive mask == fp32 reference."""
    b, hq, hkv, d, sq, lk = cfg
    torch.manual_seed(33)
    q = torch.randn(b, hq, sq, d, device="cuda").half()
    k, v = (torch.randn(b, hkv, lk, d, device="cuda").half() for _ in range(2))
    slopes = _alibi_slopes(hq).cuda()
    qi = torch.arange(sq, device="cuda")[:, None] + (lk - sq)
    ki = torch.arange(lk, device="cuda")[None]
    bias = (slopes[:, None, None] * (ki - qi).float()[None])[None].expand(b, hq, sq, lk)
    scale = d ** -0.5
    ref = _sdpa_ref(q, k, v, scale, True, mask=bias)
    out = hip.attn_prefill(q, k, v, scale=scale, causal=True, alibi=slopes)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-3)
    out = hip.attn_prefill(q, k, v, scale=scale, causal=True, mask=bias.half().contiguous())
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=4e-3)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("cfg", [(1, 32, 32, 128, 1, 1000), (2, 8, 2, 128, 3, 255), (1, 4, 4, 64, 6, 254)])
def test_attn_decode_alibi(hip, fp8, cfg):
    """Decode attention with ALiBi slopes and an unrotated fused append (no RoPE table), against fp32 math on what the
    cache holds."""
    torch.manual_seed(34)
    b, hq, hkv, d, q_len, past = cfg
    lmax = 1280
    ld = (hq + 2 * hkv) * d
    kc = (torch.randn(b, hkv, lmax, d) * 0.5).half().cuda()
    vc = (torch.randn(b, hkv, d, lmax) * 0.5).half().cuda()
    if fp8:
        kq, vq = torch.empty_like(kc, dtype=torch.uint8), torch.empty_like(vc, dtype=torch.uint8)
        hip.to_e5m2(kc, kq)
        hip.to_e5m2(vc, vq)
        kc, vc = kq, vq
    qkv = (torch.randn(b * q_len, ld) * 0.5).half().cuda()
    past_len = torch.full((b,), past, dtype=torch.int32, device="cuda")
    slopes = _alibi_slopes(hq).cuda()
    scale = 1.0 / math.sqrt(d)
    for splits in (1, 3):
        ws = torch.empty(hip.attn_decode_ws_floats(b, hq, hkv, q_len, d, splits) + 16, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(b * hkv, dtype=torch.int32, device="cuda")
        out = torch.empty(b * q_len, hq * d, dtype=torch.float16, device="cuda")
        hip.attn_decode(qkv, kc, vc, past_len, out, ws, cnt, n_heads=hq, n_kv_heads=hkv, head_dim=d, q_len=q_len,
                        batch=b, scale=scale, splits=splits, append=True, alibi=slopes)
        total = past + q_len
        kf = (kc.view(torch.float8_e5m2) if fp8 else kc).float()[:, :, :total]
        vf = (vc.view(torch.float8_e5m2) if fp8 else vc).float()[:, :, :, :total].transpose(-1, -2)
        x = qkv.float().view(b, q_len, hq + 2 * hkv, d).transpose(1, 2)
        if not fp8:   # the appended rows are the new tokens' k / v
            torch.testing.assert_close(kf[:, :, past:], x[:, hq:hq + hkv], rtol=1e-3, atol=1e-3)
        qi = torch.arange(q_len, device="cuda")[:, None] + past
        ki = torch.arange(total, device="cuda")[None]
        bias = (slopes[:, None, None] * (ki - qi).float()[None])[None]
        ref = _sdpa_ref(x[:, :hq], kf, vf, scale, True, mask=bias)
        torch.testing.assert_close(out.float().view(b, q_len, hq * d), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("cfg", [(1, 8, 2, 128, 100, 0, 128, False), (2, 4, 4, 64, 70, 0, 64, True), (1, 4, 2, 128, 37, 19, 64, False),
                                 (1, 4, 1, 96, 64, 0, 96, False)])
def test_prefill_rope_append(hip, fp8, cfg):
    b, hq, hkv, d, s, past, rot, inter = cfg
    torch.manual_seed(32)
    lmax = 256
    qkv = torch.randn(b, s, (hq + 2 * hkv) * d, device="cuda").half()
    ref = qkv.clone()
    inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, device="cuda").float() / rot))
    ang = torch.arange(lmax, device="cuda").float()[:, None] * inv[None]
    table = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
    dt = torch.uint8 if fp8 else torch.float16
    kc = torch.zeros(b, hkv, lmax, d, dtype=dt, device="cuda")
    vc = torch.zeros(b, hkv, d, lmax, dtype=dt, device="cuda")
    scratch = torch.zeros(b, hkv, d, (s + 7) // 8 * 8, dtype=torch.float16, device="cuda") if fp8 else None
    hip.prefill_rope_append(qkv, kc, vc, table, n_heads=hq, n_kv_heads=hkv, head_dim=d, past=past, rot_dim=rot,
                            interleaved=inter, vt_scratch=scratch)
    # reference: separate RoPE kernel on q / k views, plain copies into the caches
    pos = torch.arange(past, past + s, device="cuda")[None].expand(b, s)
    q_ref = ref[..., : hq * d].view(b, s, hq, d).transpose(1, 2)
    k_ref = ref[..., hq * d:(hq + hkv) * d].view(b, s, hkv, d).transpose(1, 2)
    v_ref = ref[..., (hq + hkv) * d:].view(b, s, hkv, d).transpose(1, 2)
    hip.rope_inplace(q_ref, position_ids=pos, inv_freq=inv, interleaved=inter, rot_dim=rot)
    hip.rope_inplace(k_ref, position_ids=pos, inv_freq=inv, interleaved=inter, rot_dim=rot)
    torch.testing.assert_close(qkv.float(), ref.float(), rtol=3e-3, atol=3e-3)       # q, k rotated in place; v untouched
    if fp8:
        k_deq = kc[:, :, past:past + s].view(torch.float8_e5m2).float()
        v_deq = vc[:, :, :, past:past + s].view(torch.float8_e5m2).float()
        assert ((k_deq - k_ref.float()).abs() <= 0.13 * k_ref.float().abs() + 1e-3).all()
        assert torch.equal(v_deq, v_ref.transpose(-1, -2).to(torch.float8_e5m2).float())
        assert torch.equal(scratch[..., :s], v_ref.transpose(-1, -2))
    else:
        torch.testing.assert_close(kc[:, :, past:past + s].float(), k_ref.float(), rtol=3e-3, atol=3e-3)
        assert torch.equal(vc[:, :, :, past:past + s], v_ref.transpose(-1, -2))
    assert int(kc[:, :, :past].float().abs().sum()) == 0 and int(kc[:, :, past + s:].float().abs().sum()) == 0
    assert int(vc[:, :, :, past + s:].float().abs().sum()) == 0


# ---------------------------------------------------------------------------------------------
# RWKV recurrences
# ---------------------------------------------------------------------------------------------
def test_rwkv4_wkv_and_time_shift(hip):
    from transformers.models.rwkv.modeling_rwkv import rwkv_linear_attention_cpu
    torch.manual_seed(40)
    b, t, c = 2, 37, 192
    td, tf = torch.randn(c) * 0.5, torch.randn(c) * 0.5
    k, v = torch.randn(b, t, c), torch.randn(b, t, c)
    ref, ref_state = rwkv_linear_attention_cpu(td, tf, k, v, state=None, return_state=True)
    num = torch.zeros(b, c, device="cuda")
    den = torch.zeros(b, c, device="cuda")
    mx = torch.full((b, c), -1e38, device="cuda")
    out = hip.rwkv_linear_attention_v4(td.cuda(), tf.cuda(), k.cuda(), v.cuda(), num, den, mx)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    # the (num, den, max) triple is only defined up to the shared exponent: compare the ratio and continue a step
    k2, v2 = torch.randn(b, 1, c), torch.randn(b, 1, c)
    ref2, _ = rwkv_linear_attention_cpu(td, tf, k2, v2, state=ref_state, return_state=True)
    out2 = hip.rwkv_linear_attention_v4(td.cuda(), tf.cuda(), k2.cuda(), v2.cuda(), num, den, mx)
    torch.testing.assert_close(out2.cpu(), ref2, rtol=1e-4, atol=1e-4)
    h16 = hip.rwkv_linear_attention_v4(td.cuda(), tf.cuda(), k.cuda().half(), v.cuda().half())
    torch.testing.assert_close(h16.float().cpu(), ref, rtol=2e-2, atol=2e-2)
    hidden, shifted, mix = torch.randn(b, t, c).cuda(), torch.randn(b, t, c).cuda(), torch.rand(3, c).cuda()
    ts = hip.rwkv_time_shift(hidden, shifted, mix)
    for i in range(3):
        torch.testing.assert_close(ts[i], hidden * mix[i] + shifted * (1 - mix[i]), rtol=1e-5, atol=1e-5)


def test_rwkv5_wkv(hip):
    torch.manual_seed(41)
    b, h, t, s = 2, 3, 19, 64
    r, k, v = (torch.randn(b, h, t, s) * 0.5 for _ in range(3))
    w = torch.rand(h, s) * 0.5 + 0.4
    u = torch.randn(h, s) * 0.3
    state0 = torch.randn(b, h, s, s) * 0.1
    st = state0.clone()
    ref = torch.zeros(b, t, h, s)
    for ti in range(t):
        kv = k[:, :, ti, :, None] * v[:, :, ti, None, :]                       # [b, h, i, j]
        ref[:, ti] = torch.einsum("bhi,bhij->bhj", r[:, :, ti], u[None, :, :, None] * kv + st)
        st = st * w[None, :, :, None] + kv
    dev_state = state0.clone().cuda()
    out = hip.rwkv_linear_attention_v5(w.cuda(), u.cuda(), r.cuda(), k.cuda(), v.cuda(), dev_state)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dev_state.cpu(), st, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------
# GEMV with the LayerNorm folded in
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["sym_int4", "asym_int4", "nf4", "sym_int8"])
@pytest.mark.parametrize("cfg", [(1, 96, 512, False), (3, 200, 1024, False), (5, 64, 4096, True), (2, 128, 4544, False)])
def test_gemv_layernorm_fold(hip, name, cfg):
    """rs * (W.(x*gamma)) - rs*mean*(W.gamma) + W.beta == LayerNorm(x) @ W^T, with a DC offset on x so that the mean
    term matters; plain, with bias + activation + residual, and the gated two-weight form."""
    m, n, k, gated = cfg
    qt = Q.ggml_tensor_qtype[name]
    torch.manual_seed(41)
    w, wr = _quant_dev(torch.randn(n, k) / math.sqrt(k), qt)
    x = (torch.randn(m, k) * 0.7 + 0.3).half()
    gamma, beta = (1 + 0.1 * torch.randn(k)).half(), (0.1 * torch.randn(k)).half()
    bias = (0.1 * torch.randn(n)).half()
    res = torch.randn(m, n).half()
    xn = torch.nn.functional.layer_norm(x.float(), (k,), gamma.float(), beta.float(), 1e-5)
    ln = hip.layer_norm_fold(wr.cuda(), gamma.cuda(), beta.cuda())
    got = hip.gemv_lowbit_ln(x.cuda(), w, qt, n, k, gamma.cuda(), ln, eps=1e-5)
    want = xn @ wr.float().t()
    torch.testing.assert_close(got.float().cpu(), want, rtol=2e-2, atol=2e-2)
    got = hip.gemv_lowbit_ln(x.cuda(), w, qt, n, k, gamma.cuda(), ln, bias=bias.cuda(), residual=res.cuda(), eps=1e-5,
                             act="gelu_tanh")
    want2 = torch.nn.functional.gelu(want + bias.float(), approximate="tanh") + res.float()
    torch.testing.assert_close(got.float().cpu(), want2, rtol=2e-2, atol=2e-2)
    # fp32 output (the lm_head form) and no beta
    ln0 = hip.layer_norm_fold(wr.cuda(), gamma.cuda(), None)
    out32 = torch.empty(m, n, dtype=torch.float32, device="cuda")
    hip.gemv_lowbit_ln(x.cuda(), w, qt, n, k, gamma.cuda(), ln0, eps=1e-5, out=out32)
    xn0 = torch.nn.functional.layer_norm(x.float(), (k,), gamma.float(), None, 1e-5)
    torch.testing.assert_close(out32.cpu(), xn0 @ wr.float().t(), rtol=2e-2, atol=2e-2)
    if gated:
        w2, w2r = _quant_dev(torch.randn(n, k) / math.sqrt(k), qt)
        ln2 = hip.layer_norm_fold(w2r.cuda(), gamma.cuda(), beta.cuda())
        got = hip.gemv_lowbit_ln(x.cuda(), w, qt, n, k, gamma.cuda(), ln, w2=w2, ln2=ln2, eps=1e-5, act="silu")
        want3 = torch.nn.functional.silu(want) * (xn @ w2r.float().t())
        torch.testing.assert_close(got.float().cpu(), want3, rtol=2e-2, atol=2e-2)
    assert not hip.gemv_x_in_lds(16, 11008)
    with pytest.raises(ValueError):
        hip.gemv_lowbit_ln(torch.zeros(16, 11008, dtype=torch.float16, device="cuda"), w, qt, n, 11008, gamma.cuda(), ln)


@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("cfg", [(1, 32, 32, 128, 1, 1000), (2, 32, 8, 128, 1, 300), (1, 8, 2, 64, 5, 77),
                                 (1, 4, 4, 256, 3, 513), (2, 48, 1, 128, 1, 130), (3, 8, 8, 128, 9, 40)])
def test_int4_kv_cache_append_and_decode(hip, cfg, interleaved):
    """int4 caches: the prompt-side quantizer and the in-kernel append agree with a plain per-32-channel symmetric
    quantization, and the attention output equals fp32 attention over exactly the dequantized cache contents."""
    torch.manual_seed(11)
    b, hq, hkv, d, q_len, past = cfg
    lmax = 1280
    ld = (hq + 2 * hkv) * d
    shapes = hip.kv_int4_shapes(b, hkv, lmax, d)
    kc, vc = (torch.zeros(s, dtype=torch.uint8, device="cuda") for s in shapes[:2])
    ks, vs = (torch.zeros(s, dtype=torch.float16, device="cuda") for s in shapes[2:])
    inv_freq = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    ang = torch.arange(lmax)[:, None].float() * inv_freq[None]
    table = torch.stack([ang.cos(), ang.sin()], -1).contiguous().cuda()

    def rope(x, pos):  # x [B,H,S,D]
        a = pos[None, None, :, None].float() * inv_freq
        if interleaved:
            x0, x1 = x[..., 0::2], x[..., 1::2]
            return torch.stack([x0 * a.cos() - x1 * a.sin(), x1 * a.cos() + x0 * a.sin()], -1).flatten(-2)
        x0, x1 = x[..., : d // 2], x[..., d // 2:]
        return torch.cat([x0 * a.cos() - x1 * a.sin(), x1 * a.cos() + x0 * a.sin()], -1)

    def fake_quant(t):  # [..., D] -> what an int4 cache with fp16 scales per 32 channels holds
        g = t.half().float().reshape(*t.shape[:-1], d // 32, 32)
        sc = (g.abs().amax(-1, keepdim=True) / 7).half().float()
        q = torch.clamp(torch.round(g / sc.clamp_min(1e-30)), -8, 7)
        return (q * sc).reshape(t.shape)

    # the prompt: rotated keys (as prefill_rope_append leaves them in qkv) + values -> caches
    prompt = (torch.randn(b, past, ld) * 0.5).half()
    x = prompt.float().view(b, past, hq + 2 * hkv, d).transpose(1, 2)
    k_prompt, v_prompt = x[:, hq:hq + hkv], x[:, hq + hkv:]
    hip.kv_int4_append(prompt.cuda(), kc, vc, ks, vs, n_heads=hq, n_kv_heads=hkv, head_dim=d, past=0)
    k_deq, v_deq = hip.kv_int4_dequant(kc, vc, ks, vs, past)
    def flips(a, ref):      # x * (1 / scale) vs x / scale: a value on a rounding boundary may land one step off
        return ((a - ref).abs() > 2e-3).float().mean().item()
    assert flips(k_deq.float().cpu(), fake_quant(k_prompt)) < 1e-4
    assert flips(v_deq.float().cpu().transpose(-1, -2), fake_quant(v_prompt)) < 1e-4

    # one decode call with q_len new tokens: rotation + quantized append inside the attention kernel
    qkv = (torch.randn(b * q_len, ld) * 0.5).half()
    y = qkv.float().view(b, q_len, hq + 2 * hkv, d).transpose(1, 2)
    pos = torch.arange(past, past + q_len)
    q_rot, k_new, v_new = rope(y[:, :hq], pos), rope(y[:, hq:hq + hkv], pos), y[:, hq + hkv:]
    past_len = torch.full((b,), past, dtype=torch.int32, device="cuda")
    scale = 1.0 / math.sqrt(d)
    for splits in (1, 3):
        out = torch.empty(b * q_len, hq * d, dtype=torch.float16, device="cuda")
        ws = torch.empty(hip.attn_decode_ws_floats(b, hq, hkv, q_len, d, splits) + 16, dtype=torch.float32, device="cuda")
        counters = torch.zeros(hip.attn_decode_counters(b, hq, hkv), dtype=torch.int32, device="cuda")
        hip.attn_decode(qkv.cuda(), kc, vc, past_len, out, ws, counters, n_heads=hq, n_kv_heads=hkv, head_dim=d,
                        q_len=q_len, batch=b, scale=scale, splits=splits, rope_table=table, interleaved=interleaved,
                        k_scale=ks, v_scale=vs)
        k_all, v_all = hip.kv_int4_dequant(kc, vc, ks, vs, past + q_len)
        k_all, v_all = k_all.float().cpu(), v_all.float().cpu().transpose(-1, -2)
        # the appended tokens: within half a quantization step of the true rotated key / value
        bad = ((k_all[:, :, past:] - fake_quant(k_new)).abs() > 2e-3).float().mean()
        assert bad < 0.02, bad            # table cos/sin in fp32 vs torch: a rounding flips now and then
        assert flips(v_all[:, :, past:], fake_quant(v_new)) < 1e-3
        ref = _ref_attention(q_rot.half().float(), k_all, v_all, scale, past=past)
        got = out.cpu().float().view(b, q_len, hq, d).transpose(1, 2)
        torch.testing.assert_close(got, ref, rtol=1e-2, atol=1e-2)
        assert int(counters.abs().sum()) == 0
    # and against the unquantized values: int4 noise only
    exact = _ref_attention(q_rot, torch.cat([k_prompt, k_new], 2), torch.cat([v_prompt, v_new], 2), scale, past=past)
    rel = ((got - exact).abs().mean() / exact.abs().mean()).item()
    assert rel < 0.25, rel


@pytest.mark.gpu
@pytest.mark.parametrize("d", [72, 80])
@pytest.mark.parametrize("causal,sq,lk,hq,hkv", [(False, 729, 729, 16, 16), (True, 200, 200, 8, 2), (True, 5, 133, 4, 4)])
def test_attn_prefill_head_dim_72_80(d, causal, sq, lk, hq, hkv):
    """SigLIP (72) and phi-2 (80) heads run in 96-wide tiles straight from the unpadded tensors."""
    from bigdl_amd.ops import hip
    g = torch.Generator(device="cuda").manual_seed(d + sq)
    # views into a fused [B, S, H * D] projection output, as the module path hands them over
    q = torch.randn(2, sq, hq * d, device="cuda", generator=g).half().view(2, sq, hq, d).transpose(1, 2)
    k = torch.randn(2, lk, hkv * d, device="cuda", generator=g).half().view(2, lk, hkv, d).transpose(1, 2)
    v = torch.randn(2, lk, hkv * d, device="cuda", generator=g).half().view(2, lk, hkv, d).transpose(1, 2)
    out = hip.attn_prefill(q, k, v, scale=d ** -0.5, causal=causal)
    kf = k.float().repeat_interleave(hq // hkv, 1)
    vf = v.float().repeat_interleave(hq // hkv, 1)
    s = q.float() @ kf.transpose(-1, -2) * d ** -0.5
    if causal:
        keep = torch.ones(sq, lk, dtype=torch.bool, device="cuda").tril(lk - sq)
        s = s.masked_fill(~keep, float("-inf"))
    want = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(2, sq, hq * d)
    assert out.shape == want.shape
    assert (out.float() - want).abs().max().item() < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("d,hq,hkv,sq,lk", [(128, 8, 8, 96, 1000), (64, 8, 2, 130, 387), (256, 4, 4, 17, 80)])
def test_attn_prefill_reads_the_e5m2_cache_directly(d, hq, hkv, sq, lk):
    """A later chunk of a prompt attends to the fp8 cache where it lies: same result as attention over the dequantized
    copy of those bytes."""
    from bigdl_amd.ops import hip
    g = torch.Generator(device="cuda").manual_seed(d + sq)
    cap = 1024
    q = torch.randn(2, hq, sq, d, device="cuda", generator=g).half()
    kc = torch.randn(2, hkv, cap, d, device="cuda", generator=g).half().view(torch.int16).bitwise_right_shift(8).to(torch.uint8)
    vc = torch.randn(2, hkv, d, cap, device="cuda", generator=g).half().view(torch.int16).bitwise_right_shift(8).to(torch.uint8)
    k8, v8 = kc[:, :, :lk], vc[:, :, :, :lk]
    kf_all = torch.empty(kc.shape, dtype=torch.float16, device="cuda")
    vf_all = torch.empty(vc.shape, dtype=torch.float16, device="cuda")
    hip.from_e5m2(kc, kf_all)
    hip.from_e5m2(vc, vf_all)
    kf, vf = kf_all[:, :, :lk], vf_all[:, :, :, :lk]
    want = hip.attn_prefill(q, kf, vf, scale=d ** -0.5, causal=True, v_transposed=True)
    got = hip.attn_prefill(q, k8, v8, scale=d ** -0.5, causal=True, v_transposed=True, kv_e5m2=True)
    assert torch.isfinite(got.float()).all()
    assert (got.float() - want.float()).abs().max().item() < 2e-3


@pytest.mark.parametrize("shape", [(5, 64), (300, 4096), (33, 11008)])
def test_quant_rows_e4m3_matches_torch_conversion(hip, shape):
    """row-wise activation quantization of the opt-in fp8 prompt pass: scale = max|x| / 448 per row, codes equal to
    PyTorch's own fp32 -> float8_e4m3fn conversion of x / scale"""
    torch.manual_seed(0)
    m, k = shape
    x = (torch.randn(m, k) * torch.rand(m, 1) * 3).half().cuda()
    x[0, :8] = 0
    q, sc = hip.quant_rows_e4m3(x)
    want_sc = x.float().abs().amax(1, keepdim=True) / 448.0
    torch.testing.assert_close(sc, want_sc, rtol=1e-6, atol=0)
    want = (x.float() / sc).to(torch.float8_e4m3fn)
    same = (q.view(torch.uint8) == want.view(torch.uint8)).float().mean().item()
    assert same > 0.999, same                   # (a product that lands exactly between two codes may round the other way)
    back = q.float() * sc
    assert ((back - x.float()).abs().mean() / x.float().abs().mean()).item() < 0.04


def test_bf16_to_f16(hip):
    torch.manual_seed(0)
    x = (torch.randn(37, 1000) * 100).to(torch.bfloat16).cuda()
    x[0, 0], x[0, 1] = 1.0e6, -1.0e6
    want = x.float().clamp(-65504, 65504).half()
    assert torch.equal(hip.bf16_to_f16(x), want)


@pytest.mark.parametrize("rows,v,dtype", [(1, 32000, torch.float32), (5, 128256, torch.float32), (3, 32000, torch.float16),
                                          (70, 9000, torch.float32), (2, 1003, torch.float32)])
def test_argmax_split_rows_and_fused_increment(hip, rows, v, dtype):
    """Long rows of few-row calls are reduced by 8 workgroups and a last-arriver; ties resolve to the lowest index; the
    launch can also advance two int32 vectors (cache lengths of the decode graph)."""
    torch.manual_seed(3)
    x = torch.randn(rows, (v + 7) // 8 * 8, device="cuda").to(dtype)[:, :v]
    x[0, v // 3] = 50.0
    x[0, v - 2] = 50.0                      # a tie: the first one wins
    x[rows - 1, v - 1] = 60.0               # winner in the scalar tail / last chunk
    out = torch.empty(rows, dtype=torch.int64, device="cuda")
    dst = torch.arange(6, dtype=torch.int32, device="cuda")
    src = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.int32, device="cuda")
    for _ in range(3):                      # tickets must be left ready for the next launch
        hip.argmax(x, out=out, inc=(dst, src, 2))
    ref = x.float().cpu().argmax(-1)
    if rows > 1:
        assert out[0].item() == v // 3
    assert torch.equal(out.cpu()[1:rows - 1], ref[1:rows - 1])
    assert out[rows - 1].item() == v - 1
    assert dst.cpu().tolist() == [0 + 6, 1, 2 + 6, 3 + 6, 4, 5 + 6]
    assert torch.equal(hip.argmax(x).cpu(), out.cpu())
