"""Host quantizer / dequantizer / re-layout checks against plain PyTorch oracles."""
import numpy as np
import pytest
import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.ops import host

BLOCK_FORMATS = ["sym_int4", "asym_int4", "sym_int5", "asym_int5", "sym_int8", "nf4", "nf3", "fp4", "fp6",
                 "fp8_e4m3", "fp8_e5m2", "woq_int4", "q2_k", "q4_k", "q5_k", "q6_k", "fp6_k",
                 "gguf_iq2_xxs", "gguf_iq2_xs", "gguf_iq1_s"]
# loose upper bounds on relative RMS error for gaussian weights
REL_RMS = {"sym_int4": 0.12, "asym_int4": 0.10, "sym_int5": 0.06, "asym_int5": 0.05, "sym_int8": 0.01,
           "nf4": 0.11, "nf3": 0.22, "fp4": 0.17, "fp6": 0.06, "fp8_e4m3": 0.04, "fp8_e5m2": 0.08,
           "woq_int4": 0.13, "q2_k": 0.40, "q4_k": 0.09, "q5_k": 0.05, "q6_k": 0.025, "fp6_k": 0.07,
           "gguf_iq2_xxs": 0.55, "gguf_iq2_xs": 0.50, "gguf_iq1_s": 0.75}


def _rel_rms(a, b):
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


@pytest.mark.parametrize("name", BLOCK_FORMATS)
def test_roundtrip_error(name):
    torch.manual_seed(0)
    qt = Q.ggml_tensor_qtype[name]
    w = torch.randn(48, 512)
    buf = host.quantize(w, qt)
    assert buf.numel() * buf.element_size() == 48 * 512 // host.ggml_qk_size(qt) * host.ggml_type_size(qt)
    deq = host.dequantize(buf.view(torch.uint8) if buf.dtype != torch.uint8 else buf, qt, w.shape)
    err = _rel_rms(deq, w)
    assert err < REL_RMS[name], (name, err)


def test_block_geometry():
    assert host.ggml_qk_size(Q.SYM_INT4) == 32 and host.ggml_type_size(Q.SYM_INT4) == 18
    assert host.ggml_type_size(Q.ASYM_INT4) == 20
    assert host.ggml_type_size(Q.SYM_INT5) == 22
    assert host.ggml_type_size(Q.ASYM_INT5) == 24
    assert host.ggml_type_size(Q.SYM_INT8) == 34
    assert host.ggml_qk_size(Q.Q4_K) == 256 and host.ggml_type_size(Q.Q4_K) == 144
    assert host.ggml_type_size(Q.Q6_K) == 210 and host.ggml_type_size(Q.Q5_K) == 176
    assert host.ggml_type_size(Q.Q2_K) == 84
    assert host.ggml_type_size(Q.IQ2_XXS) == 66 and host.ggml_type_size(Q.IQ2_XS) == 74
    assert host.ggml_type_size(Q.IQ1_S) == 50


def test_q4_0_matches_python_oracle():
    """Bit layout of sym_int4 == public ggml q4_0 (value (q-8)*d, low nibble = elt j, high = j+16)."""
    torch.manual_seed(1)
    w = torch.randn(4, 64)
    buf = host.quantize(w, Q.SYM_INT4).numpy().reshape(-1, 18)
    d = buf[:, :2].copy().view(np.float16).astype(np.float32)
    qs = buf[:, 2:]
    lo = (qs & 15).astype(np.float32) - 8
    hi = (qs >> 4).astype(np.float32) - 8
    ref = np.concatenate([lo, hi], axis=1) * d
    deq = host.dequantize(torch.from_numpy(buf.reshape(-1)), Q.SYM_INT4, (4, 64)).numpy().reshape(-1, 32)
    np.testing.assert_array_equal(ref, deq)
    # the quantizer itself: d = signed-absmax / -8
    wb = w.numpy().reshape(-1, 32)
    idx = np.abs(wb).argmax(axis=1)
    dref = (wb[np.arange(len(idx)), idx] / -8).astype(np.float16).astype(np.float32)
    np.testing.assert_array_equal(d[:, 0], dref)


def test_q8_0_exact_for_integers():
    w = torch.randint(-127, 128, (2, 64)).float()
    w[:, 0] = 127
    w[:, 32] = -127
    deq = host.dequantize(host.quantize(w, Q.SYM_INT8), Q.SYM_INT8, w.shape)
    assert torch.equal(deq, w)


def test_fp8_tables_against_torch():
    codes = torch.arange(256, dtype=torch.uint8)
    for name, dt in (("fp8_e4m3", torch.float8_e4m3fn), ("fp8_e5m2", torch.float8_e5m2)):
        ref = codes.view(dt).float()
        # encode every finite fp8 value with scale 1 (block max = fp8 max -> d == 1)
        finite = torch.isfinite(ref)
        vals = ref[finite]
        fmax = vals.abs().max()
        n = (vals.numel() + 30) // 31 * 32
        w = torch.zeros(n)
        # each block of 32 gets the format max in slot 0 so that the block scale is exactly 1
        slots = torch.arange(n).reshape(-1, 32)[:, 1:].reshape(-1)[: vals.numel()]
        w[slots] = vals
        w[::32] = fmax
        qt = Q.ggml_tensor_qtype[name]
        deq = host.dequantize(host.quantize(w.reshape(1, -1), qt), qt, (1, n)).reshape(-1)
        assert torch.equal(deq[slots], vals), name


@pytest.mark.parametrize("name", ["sym_int4", "asym_int4", "fp4", "nf4", "woq_int4", "sym_int5", "asym_int5", "fp6",
                                  "sym_int8", "fp8_e4m3", "fp8_e5m2"])
@pytest.mark.parametrize("shape", [(32, 256), (40, 384), (7, 128)])
def test_relayout_roundtrip(name, shape):
    torch.manual_seed(2)
    qt = Q.ggml_tensor_qtype[name]
    n, k = shape
    if k % host.ggml_qk_size(qt):
        pytest.skip("k not a block multiple")
    w = torch.randn(n, k)
    buf = host.quantize(w, qt)
    raw = buf.view(torch.uint8) if buf.dtype != torch.uint8 else buf
    dev = host.relayout_to_device(raw, qt, n, k)
    assert dev.numel() == host.device_nbytes(qt, n, k)
    back = host.relayout_to_host(dev, qt, n, k)
    back = back.view(torch.uint8) if back.dtype != torch.uint8 else back
    assert torch.equal(back, raw)


def test_t16_layout_documented_positions():
    """Spot-check the tiled layout contract used by the HIP kernels (4-bit plane + scale plane)."""
    n, k = 16, 128
    w = torch.zeros(n, k)
    # make every block's absmax -8 so that d == 1 and code == value + 8
    vals = torch.randint(-7, 8, (n, k)).float()
    vals[:, ::32] = -8
    buf = host.quantize(vals, Q.SYM_INT4)
    dev = host.relayout_to_device(buf, Q.SYM_INT4, n, k).numpy()
    words = dev[: n * k // 2].view(np.uint32).reshape(64, 4)  # [lane][s]
    P4 = [0, 4, 1, 5, 2, 6, 3, 7]
    for lane in (0, 5, 17, 63):
        r, g = lane % 16, lane // 16
        for s in range(4):
            for j in range(8):
                code = (int(words[lane, s]) >> (4 * P4[j])) & 15
                assert code == int(vals[r, 32 * s + 8 * g + j]) + 8
    scales = dev[n * k // 2:].view(np.float16).reshape(16, 4)
    assert (scales == 1).all()


def test_matmul_q_matches_dequant_reference():
    torch.manual_seed(3)
    w = torch.randn(96, 256)
    x = torch.randn(5, 256)
    for name in ("sym_int4", "q4_k", "nf4", "fp8_e5m2"):
        qt = Q.ggml_tensor_qtype[name]
        buf = host.quantize(w, qt)
        raw = buf.view(torch.uint8) if buf.dtype != torch.uint8 else buf
        ref = x @ host.dequantize(raw, qt, w.shape).t()
        out = host.matmul_q(x, raw, qt, 96, 256)
        torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_rtn_formats():
    torch.manual_seed(4)
    w = torch.randn(8, 128)
    for name, tol in (("sym_int8_rtn", 0.02), ("sym_int4_rtn", 0.25), ("asym_int4_rtn", 0.2)):
        qt = Q.ggml_tensor_qtype[name]
        codes, scale = host.quantize(w, qt)
        assert scale.dtype == torch.float16
        deq = host.dequantize_rtn(codes, scale.float(), qt, w.shape)
        assert _rel_rms(deq, w) < tol, name


def test_imatrix_shifts_error_to_unimportant_columns():
    torch.manual_seed(5)
    w = torch.randn(16, 256)
    im = torch.ones(256)
    im[:32] = 1000.0
    for name in ("q4_k", "q2_k"):
        qt = Q.ggml_tensor_qtype[name]
        plain = host.dequantize(host.quantize(w, qt), qt, w.shape)
        weighted = host.dequantize(host.quantize(w, qt, imatrix=im), qt, w.shape)
        e_plain = (plain[:, :32] - w[:, :32]).pow(2).mean()
        e_w = (weighted[:, :32] - w[:, :32]).pow(2).mean()
        assert e_w <= e_plain * 1.05, name


def test_moe_indexes():
    sel = torch.tensor([[0, 2], [1, 2], [0, 1]], dtype=torch.int32)
    offsets, perm = host.moe_indexes(sel, 4)
    assert offsets.tolist() == [0, 2, 4, 6, 6]
    assert perm.tolist() == [0, 4, 2, 5, 1, 3]


# ---------------------------------------------------------------------------------------------------------------
# Independent oracles: the public ggml block formulas written out in numpy (the reference keeps the same formulas in
# python/llm/src/ipex_llm/transformers/gguf/gguf.py:285-367), applied to the host-layout bytes our quantizer wrote.
# ---------------------------------------------------------------------------------------------------------------
def _f16(b):
    return b.copy().view(np.float16).astype(np.float32)


def _ggml_q4_1(buf):
    b = buf.reshape(-1, 20)
    d, m, qs = _f16(b[:, 0:2]), _f16(b[:, 2:4]), b[:, 4:]
    q = np.concatenate([qs & 15, qs >> 4], axis=1).astype(np.float32)
    return q * d + m


def _ggml_q5_0(buf):
    b = buf.reshape(-1, 22)
    d, qh, qs = _f16(b[:, 0:2]), b[:, 2:6].copy().view(np.uint32), b[:, 6:]
    j = np.arange(16, dtype=np.uint32)
    lo = (qs & 15) | (((qh >> j) & 1) << 4).astype(np.uint8)
    hi = (qs >> 4) | (((qh >> (j + 16)) & 1) << 4).astype(np.uint8)
    return (np.concatenate([lo, hi], axis=1).astype(np.float32) - 16) * d


def _ggml_q5_1(buf):
    b = buf.reshape(-1, 24)
    d, m, qh, qs = _f16(b[:, 0:2]), _f16(b[:, 2:4]), b[:, 4:8].copy().view(np.uint32), b[:, 8:]
    j = np.arange(16, dtype=np.uint32)
    lo = (qs & 15) | (((qh >> j) & 1) << 4).astype(np.uint8)
    hi = (qs >> 4) | (((qh >> (j + 16)) & 1) << 4).astype(np.uint8)
    return np.concatenate([lo, hi], axis=1).astype(np.float32) * d + m


def _ggml_q8_0(buf):
    b = buf.reshape(-1, 34)
    return b[:, 2:].copy().view(np.int8).astype(np.float32) * _f16(b[:, 0:2])


def _ggml_q6_k(buf):
    b = buf.reshape(-1, 210)
    ql, qh, sc, d = b[:, :128], b[:, 128:192], b[:, 192:208].copy().view(np.int8).astype(np.float32), _f16(b[:, 208:210])
    out = np.zeros((b.shape[0], 256), dtype=np.float32)
    for half in range(2):
        l_ = ql[:, 64 * half: 64 * half + 64].astype(np.int32)
        h_ = qh[:, 32 * half: 32 * half + 32].astype(np.int32)
        s_ = sc[:, 8 * half: 8 * half + 8]
        pos = np.arange(32)
        isc = pos // 16
        q1 = ((l_[:, :32] & 15) | (((h_ >> 0) & 3) << 4)) - 32
        q2 = ((l_[:, 32:] & 15) | (((h_ >> 2) & 3) << 4)) - 32
        q3 = ((l_[:, :32] >> 4) | (((h_ >> 4) & 3) << 4)) - 32
        q4 = ((l_[:, 32:] >> 4) | (((h_ >> 6) & 3) << 4)) - 32
        base = 128 * half
        out[:, base: base + 32] = d * s_[:, isc] * q1
        out[:, base + 32: base + 64] = d * s_[:, isc + 2] * q2
        out[:, base + 64: base + 96] = d * s_[:, isc + 4] * q3
        out[:, base + 96: base + 128] = d * s_[:, isc + 6] * q4
    return out


@pytest.mark.parametrize("name,oracle,qk", [("asym_int4", _ggml_q4_1, 32), ("sym_int5", _ggml_q5_0, 32),
                                            ("asym_int5", _ggml_q5_1, 32), ("sym_int8", _ggml_q8_0, 32),
                                            ("q6_k", _ggml_q6_k, 256)])
def test_block_layout_matches_public_ggml_formula(name, oracle, qk):
    torch.manual_seed(11)
    qt = Q.ggml_tensor_qtype[name]
    w = torch.randn(6, 512) * 0.3
    buf = host.quantize(w, qt)
    raw = (buf.view(torch.uint8) if buf.dtype != torch.uint8 else buf).numpy().reshape(-1)
    ref = oracle(raw).reshape(6, 512)
    deq = host.dequantize(torch.from_numpy(raw.copy()), qt, (6, 512)).numpy()
    np.testing.assert_allclose(deq, ref, rtol=1e-6, atol=1e-7)
    # and the codes are a sensible quantization of w, not merely self-consistent
    assert _rel_rms(torch.from_numpy(ref), w) < REL_RMS[name]
