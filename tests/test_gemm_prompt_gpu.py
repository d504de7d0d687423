"""The prompt-sized fused low-bit GEMM (csrc/hip/gemm_prompt_kernel.h) against the fp32 product with the dequantized
weight: every tile configuration, every tiled format, the four Llama-2-7B projection shapes, partial row / column
tiles, K splits, and the bias / activation / residual / output-type epilogues."""
import pytest
import torch

from bigdl_amd import qtypes as Q

pytestmark = pytest.mark.gpu

T16 = ["sym_int4", "asym_int4", "woq_int4", "fp4", "nf4", "sym_int5", "asym_int5", "fp6", "sym_int8", "fp8_e4m3",
       "fp8_e5m2"]


@pytest.fixture(scope="module")
def hip():
    from bigdl_amd.ops import hip as h
    h.lib()
    yield h
    h.gemm_prompt_force()


def _case(hip, qname, m, n, k, seed=5):
    from bigdl_amd.models.decoder import _random_t16
    qt = Q.ggml_tensor_qtype[qname]
    w = _random_t16(n, k, qt, torch.device("cuda"), torch.Generator(device="cuda").manual_seed(seed), scale=0.02).data
    wd = hip.dequant_t16(w, qt, n, k).float()
    x = torch.randn(m, k, device="cuda", dtype=torch.float16, generator=torch.Generator(device="cuda").manual_seed(seed + 1))
    return qt, w, x, x.float() @ wd.t()


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("n,k", [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)])
def test_llama2_7b_shapes_every_tile_configuration(hip, cfg, n, k):
    qt, w, x, ref = _case(hip, "sym_int4", 1024, n, k)
    hip.gemm_prompt_force(cfg, 0)
    got = hip.gemm_prompt(x, w, qt, n, k).float()
    hip.gemm_prompt_force()
    assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item()


def test_planned_configuration_matches_too(hip):
    for n, k in [(12288, 4096), (4096, 11008)]:
        qt, w, x, ref = _case(hip, "sym_int4", 1024, n, k)
        got = hip.gemm_prompt(x, w, qt, n, k).float()
        assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
        cfg, sk = hip.gemm_prompt_plan(1024, n, k)
        assert 0 <= cfg <= 2 and sk >= 1


@pytest.mark.parametrize("qname", T16)
@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_every_format_partial_tiles_and_epilogues(hip, qname, cfg):
    # 300 rows (partial row block), 1000 columns (partial column block, not a multiple of 16), K = 1024
    m, n, k = 300, 1000, 1024
    qt, w, x, ref = _case(hip, qname, m, n, k)
    hip.gemm_prompt_force(cfg, 0)
    try:
        got = hip.gemm_prompt(x, w, qt, n, k).float()
        tol = 2e-3 * ref.abs().max().item()
        assert (got - ref).abs().max().item() < tol
        bias = torch.randn(n, device="cuda", dtype=torch.float16)
        res = torch.randn(m, n, device="cuda", dtype=torch.float16)
        got2 = hip.gemm_prompt(x, w, qt, n, k, bias=bias, residual=res, act="silu").float()
        # the staged epilogue rounds (product + bias) to fp16 before the activation, like a separate activation kernel
        ref2 = torch.nn.functional.silu((ref + bias.float()).half().float()) + res.float()
        assert (got2 - ref2).abs().max().item() < 2e-3 * ref2.abs().max().item() + 2e-2
        got3 = hip.gemm_prompt(x, w, qt, n, k, out_dtype=torch.float32)
        assert (got3 - ref).abs().max().item() < tol
        got4 = hip.gemm_prompt(x, w, qt, n, k, bias=bias, act="gelu", out_dtype=torch.bfloat16).float()
        ref4 = torch.nn.functional.gelu((ref + bias.float()).bfloat16().float())
        assert (got4 - ref4).abs().max().item() < 1e-2 * ref4.abs().max().item() + 2e-2
    finally:
        hip.gemm_prompt_force()


@pytest.mark.parametrize("sk", [2, 3])
def test_k_splits_and_unaligned_columns(hip, sk):
    # N = 1004: not a multiple of 8, so the output goes through the fp32 slab + reduce kernel even without a K split
    for n in (1004, 1024):
        qt, w, x, ref = _case(hip, "sym_int4", 520, n, 4096)
        bias = torch.randn(n, device="cuda", dtype=torch.float16)
        for force in ((0, sk), (2, sk), (0, 1)):
            hip.gemm_prompt_force(*force)
            got = hip.gemm_prompt(x, w, qt, n, 4096, bias=bias, act="relu").float()
            hip.gemm_prompt_force()
            ref2 = torch.relu(ref + bias.float())
            assert (got - ref2).abs().max().item() < 2e-3 * ref2.abs().max().item() + 1e-2, (n, force)


def test_rows_of_a_larger_buffer(hip):
    # x rows with a stride (a slice of wider activations) and an output view with a stride
    qt, w, x, ref = _case(hip, "sym_int4", 640, 512, 1024)
    wide = torch.zeros(640, 2048, device="cuda", dtype=torch.float16)
    wide[:, 512:1536] = x
    outw = torch.zeros(640, 1024, device="cuda", dtype=torch.float16)
    hip.gemm_prompt(wide[:, 512:1536], w, qt, 512, 1024, out=outw[:, 256:768])
    assert (outw[:, 256:768].float() - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    assert outw[:, :256].abs().max().item() == 0 and outw[:, 768:].abs().max().item() == 0


@pytest.mark.parametrize("qname", ["sym_int4", "asym_int4", "nf4", "sym_int8"])
@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_gate_up_pair_in_one_launch(hip, qname, cfg):
    """act(x Wg^T + bg) * (x Wu^T + bu) from one launch against the two separate products (partial row and column blocks)."""
    from bigdl_amd.models.decoder import _random_t16
    m, n, k = 700, 1104, 1024
    qt = Q.ggml_tensor_qtype[qname]
    dev = torch.device("cuda")
    wg = _random_t16(n, k, qt, dev, torch.Generator(device="cuda").manual_seed(1), scale=0.02).data
    wu = _random_t16(n, k, qt, dev, torch.Generator(device="cuda").manual_seed(2), scale=0.02).data
    x = torch.randn(m, k, device="cuda", dtype=torch.float16)
    bg = torch.randn(n, device="cuda", dtype=torch.float16) * 0.1
    bu = torch.randn(n, device="cuda", dtype=torch.float16) * 0.1
    g = (x.float() @ hip.dequant_t16(wg, qt, n, k).float().t() + bg.float()).half().float()
    u = (x.float() @ hip.dequant_t16(wu, qt, n, k).float().t() + bu.float()).half().float()
    ref = torch.nn.functional.silu(g) * u
    hip.gemm_prompt_force(cfg, 0)
    try:
        got = hip.gemm_prompt_pair(x, wg, wu, qt, n, k, bias=bg, bias2=bu, act="silu").float()
    finally:
        hip.gemm_prompt_force()
    assert (got - ref).abs().max().item() < 4e-3 * ref.abs().max().item() + 2e-3
    # Llama-2-7B shape, planned configuration, no biases
    n, k = 11008, 4096
    wg = _random_t16(n, k, Q.SYM_INT4, dev, torch.Generator(device="cuda").manual_seed(3), scale=0.02).data
    wu = _random_t16(n, k, Q.SYM_INT4, dev, torch.Generator(device="cuda").manual_seed(4), scale=0.02).data
    x = torch.randn(1024, k, device="cuda", dtype=torch.float16)
    if qname == "sym_int4" and cfg == 0:
        g = hip.gemm_prompt(x, wg, Q.SYM_INT4, n, k).float()
        u = hip.gemm_prompt(x, wu, Q.SYM_INT4, n, k).float()
        ref = torch.nn.functional.silu(g) * u
        got = hip.gemm_prompt_pair(x, wg, wu, Q.SYM_INT4, n, k).float()
        assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3
