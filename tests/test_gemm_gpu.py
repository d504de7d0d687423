"""Dequant-fused MFMA GEMM (dense and grouped) against a plain PyTorch fp32 reference over the dequantized weights."""
import pytest
import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.ops import host

pytestmark = pytest.mark.gpu

T16 = ["sym_int4", "asym_int4", "woq_int4", "fp4", "nf4", "sym_int5", "asym_int5", "fp6", "sym_int8", "fp8_e4m3",
       "fp8_e5m2"]


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


@pytest.mark.parametrize("name", T16)
@pytest.mark.parametrize("m", [17, 64, 100, 128, 300])
def test_gemm_matches_dequant_reference(hip, name, m):
    torch.manual_seed(2)
    qt = Q.ggml_tensor_qtype[name]
    n, k = 272, 640          # N not a multiple of 128 (tail tiles), K = 5 k-tiles (odd: both pipeline exits)
    w = torch.randn(n, k) * 0.05
    x = torch.randn(m, k).half()
    dev, ref_w = _quant_dev(w, qt)
    ref = x.float() @ ref_w.t()
    out = hip.gemm_lowbit(x.cuda(), dev, qt, n, k, out_dtype=torch.float32).cpu()
    # weights are rounded to fp16 after scaling (as in dequantize + dense GEMM), accumulation is fp32
    torch.testing.assert_close(out, ref, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("shape", [(96, 4096, 1024), (130, 256, 4096), (64, 1024, 128)])
def test_gemm_split_k_epilogue(hip, shape):
    """Shapes that take the split-K path (few output tiles, long K) with bias, activation and residual."""
    torch.manual_seed(3)
    m, k, n = shape
    qt = Q.SYM_INT4
    w = torch.randn(n, k) * 0.03
    x = torch.randn(m, k).half()
    bias = (torch.randn(n) * 0.1).half()
    res = torch.randn(m, n).half()
    dev, ref_w = _quant_dev(w, qt)
    ref = torch.nn.functional.silu(x.float() @ ref_w.t() + bias.float()) + res.float()
    out = hip.gemm_lowbit(x.cuda(), dev, qt, n, k, bias=bias.cuda(), residual=res.cuda(), act="silu").cpu().float()
    torch.testing.assert_close(out, ref, rtol=5e-3, atol=5e-3)


def test_gemm_strided_rows_and_k_tail(hip):
    torch.manual_seed(4)
    qt = Q.ASYM_INT4
    m, k, n = 70, 160, 48        # K = 1.25 k-tiles: zero-scale padding blocks, x beyond K must not be read
    w = torch.randn(n, k) * 0.05
    big = torch.randn(m, k + 24).half().cuda()
    x = big[:, :k]
    dev, ref_w = _quant_dev(w, qt)
    out = hip.gemm_lowbit(x, dev, qt, n, k, out_dtype=torch.float32).cpu()
    torch.testing.assert_close(out, x.cpu().float() @ ref_w.t(), rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("bm", [64, 128])
def test_grouped_gemm_matches_per_expert_reference(hip, bm):
    torch.manual_seed(5)
    qt = Q.SYM_INT4
    e, n, k, t, top_k = 6, 96, 256, 150, 2
    ws = [torch.randn(n, k) * 0.05 for _ in range(e)]
    refs, raws = [], []
    for w in ws:
        buf = host.quantize(w, qt)
        raws.append(buf.view(torch.uint8))
        refs.append(host.dequantize(raws[-1], qt, w.shape))
    stacked = host.relayout_to_device(torch.cat(raws), qt, e * n, k).cuda()
    x = torch.randn(t, k).half().cuda()
    ids = torch.stack([torch.randperm(e)[:top_k] for _ in range(t)]).int().cuda()
    ids[ids == 4] = 1                         # expert 4 gets no rows at all
    flat = ids.view(-1).long()
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=e)
    tok = torch.div(order, top_k, rounding_mode="floor").int()
    scale = torch.rand(t * top_k).cuda()
    rows = t * top_k
    table = hip.moe_block_table(counts, bm, (rows + bm - 1) // bm + e)
    out = hip.gemm_lowbit_grouped(x, stacked, qt, e, n, k, table, rows, bm, x_rows=tok, row_scale=scale[order].contiguous())
    exp_sorted = flat[order].cpu()
    ref = torch.stack([x[tok[i]].cpu().float() @ refs[int(exp_sorted[i])].t() * float(scale[order][i]) for i in range(rows)])
    torch.testing.assert_close(out.cpu().float(), ref, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("shape", [(64, 1024, 384), (24, 512, 2048), (100, 4096, 512)])     # (rows, K, N)
@pytest.mark.parametrize("name", ["sym_int4", "asym_int4", "fp8_e5m2"])
def test_gemm_ex_folded_norm_residual_and_row_sums(hip, name, shape):
    """RMSNorm folded into the product (gamma at x staging, 1 / rms from row sums of squares), residual add, and the
    sums of squares of the rows written, against fp32 PyTorch."""
    torch.manual_seed(5)
    m, k, n = shape
    qt = Q.ggml_tensor_qtype[name]
    w = torch.randn(n, k) * 0.03
    x = torch.randn(m, k).half()
    gamma = (1.0 + 0.1 * torch.randn(k)).half()
    bias = (torch.randn(n) * 0.1).half()
    res = torch.randn(m, n).half()
    dev, ref_w = _quant_dev(w, qt)
    eps = 1e-5
    xn = x.float() * torch.rsqrt(x.float().square().mean(-1, keepdim=True) + eps) * gamma.float()
    ref = xn @ ref_w.t() + bias.float() + res.float()
    ssq_in = x.float().square().sum(-1).cuda()
    ssq_out = torch.zeros(m, dtype=torch.float32, device="cuda")
    out = hip.gemm_lowbit_ex(x.cuda(), dev, qt, n, k, bias=bias.cuda(), residual=res.cuda(), gamma=gamma.cuda(), eps=eps,
                             row_ssq=ssq_in, ssq_out=ssq_out)
    torch.testing.assert_close(out.cpu().float(), ref, rtol=6e-3, atol=6e-3)
    torch.testing.assert_close(ssq_out.cpu(), out.cpu().float().square().sum(-1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("m", [20, 64, 130])
def test_gemm_ex_gate_up_pair(hip, m):
    """gate and up in one launch: act(norm(x) Wg^T) * (norm(x) Wu^T), with the folded norm's gemma-style offset"""
    torch.manual_seed(6)
    k, n = 768, 1408
    qt = Q.SYM_INT4
    wg, wu = torch.randn(n, k) * 0.03, torch.randn(n, k) * 0.03
    x = torch.randn(m, k).half()
    gamma = (0.1 * torch.randn(k)).half()
    dg, rg = _quant_dev(wg, qt)
    du, ru = _quant_dev(wu, qt)
    eps = 1e-6
    xn = x.float() * torch.rsqrt(x.float().square().mean(-1, keepdim=True) + eps) * (1.0 + gamma.float())
    ref = torch.nn.functional.silu(xn @ rg.t()) * (xn @ ru.t())
    out = hip.gemm_lowbit_ex(x.cuda(), dg, qt, n, k, w2=du, gamma=gamma.cuda(), gamma_offset=1.0, eps=eps,
                             row_ssq=x.float().square().sum(-1).cuda(), act="silu")
    torch.testing.assert_close(out.cpu().float(), ref, rtol=8e-3, atol=8e-3)


def _random_dev_weight(n, k, qt, seed):
    """Random packed weight in device layout plus its fp32 expansion (built on the GPU: host quantization of a
    production-sized matrix would dominate the test)."""
    from bigdl_amd.models.decoder import _random_t16
    from bigdl_amd.transformers.low_bit_linear import dequant_weight
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    p = _random_t16(n, k, qt, torch.device("cuda"), gen, scale=1.0 / k ** 0.5)
    return p, dequant_weight(p).float()


@pytest.mark.parametrize("shape", [(12288, 4096), (4096, 11008), (4096, 4096), (22016, 4096)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("m", [24, 64, 256])
def test_gemm_at_serving_shapes(hip, m, shape):
    """The 17..256-row kernel at the Llama-2-7B products of a batched decode / chunked prompt step."""
    n, k = shape
    qt = Q.SYM_INT4
    p, ref_w = _random_dev_weight(n, k, qt, 11)
    torch.manual_seed(m)
    x = torch.randn(m, k, device="cuda").half()
    out = hip.gemm_lowbit(x, p.data, qt, n, k, out_dtype=torch.float32)
    ref = x.float() @ ref_w.t()
    torch.testing.assert_close(out, ref, rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("rows_total", [64, 512, 2048])
def test_grouped_gemm_at_mixtral_shapes(hip, rows_total):
    """E = 8 experts of 14336 x 4096 (w1/w3) and 4096 x 14336 (w2), top-2 routing, 64..2048 routed rows."""
    e, top_k, bm = 8, 2, 64
    qt = Q.SYM_INT4
    t = rows_total // top_k
    torch.manual_seed(rows_total)
    for n, k in ((14336, 4096), (4096, 14336)):
        p, ref_w = _random_dev_weight(e * n, k, qt, 13)
        x = torch.randn(t, k, device="cuda").half()
        ids = torch.stack([torch.randperm(e)[:top_k] for _ in range(t)]).int().cuda()
        flat = ids.view(-1).long()
        order = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=e)
        tok = torch.div(order, top_k, rounding_mode="floor").int()
        scale = torch.rand(t * top_k, device="cuda")
        rows = t * top_k
        table = hip.moe_block_table(counts, bm, (rows + bm - 1) // bm + e)
        out = hip.gemm_lowbit_grouped(x, p.data, qt, e, n, k, table, rows, bm, x_rows=tok, row_scale=scale[order].contiguous())
        experts = flat[order]
        ref = torch.empty(rows, n, device="cuda")
        for j in range(e):
            sel = (experts == j).nonzero().squeeze(1)
            if sel.numel():
                ref[sel] = (x[tok[sel].long()].float() @ ref_w[j * n:(j + 1) * n].t()) * scale[order][sel, None]
        torch.testing.assert_close(out.float(), ref, rtol=6e-3, atol=6e-3)
        del p, ref_w, ref, out
