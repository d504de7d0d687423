"""Quantization-format registry.

Names and integer ids are the persisted contract (``config.json["bigdl_transformers_low_bit"]``
stores the *name*); they match the reference table in
python/llm/src/ipex_llm/ggml/quantize.py:28-61.
"""
from __future__ import annotations

ggml_tensor_qtype = {
    "sym_int4": 2,        # ggml q4_0
    "asym_int4": 3,       # ggml q4_1
    "sym_int5": 6,        # ggml q5_0
    "asym_int5": 7,       # ggml q5_1
    "sym_int8": 8,        # ggml q8_0
    "nf4": 10,
    "nf3": 11,
    "fp16": 12,
    "fp8_e4m3": 15,
    "fp4": 16,
    "mixed_fp4": 17,      # per-tensor choice between sym_int4 and fp4 by dequant MSE
    "mixed_fp8": 18,      # per-tensor choice between sym_int8 and fp8_e4m3 by dequant MSE
    "fp8_e5m2": 19,
    "fp8": 19,
    "bf16": 20,
    "gguf_iq2_xxs": 21,
    "gguf_iq2_xs": 22,
    "q2_k": 23,
    "gguf_iq1_s": 24,
    "gguf_iq1_m": 25,
    "q6_k": 26,
    "q4_k": 27,
    "q5_k": 28,
    "fp6": 29,
    "fp6_k": 30,
    "sym_int4_rtn": 31,
    "sym_int8_rtn": 32,
    "asym_int4_rtn": 33,
    "woq_int4": 34,
}

# llama.cpp style mixed-precision presets (resolved per tensor by
# bigdl_amd.transformers.utils.get_cur_qtype_and_imatrix)
gguf_mixed_qtype = {"gguf_q4k_s": 101, "gguf_q4k_m": 102}

_id_to_name = {}
for _k, _v in ggml_tensor_qtype.items():
    _id_to_name.setdefault(_v, _k)

SYM_INT4 = ggml_tensor_qtype["sym_int4"]
ASYM_INT4 = ggml_tensor_qtype["asym_int4"]
SYM_INT5 = ggml_tensor_qtype["sym_int5"]
ASYM_INT5 = ggml_tensor_qtype["asym_int5"]
SYM_INT8 = ggml_tensor_qtype["sym_int8"]
NF4 = ggml_tensor_qtype["nf4"]
NF3 = ggml_tensor_qtype["nf3"]
FP16 = ggml_tensor_qtype["fp16"]
BF16 = ggml_tensor_qtype["bf16"]
FP8E4 = ggml_tensor_qtype["fp8_e4m3"]
FP8E5 = ggml_tensor_qtype["fp8_e5m2"]
FP4 = ggml_tensor_qtype["fp4"]
FP6 = ggml_tensor_qtype["fp6"]
MOFQ4 = ggml_tensor_qtype["mixed_fp4"]
MOFQ8 = ggml_tensor_qtype["mixed_fp8"]
IQ2_XXS = ggml_tensor_qtype["gguf_iq2_xxs"]
IQ2_XS = ggml_tensor_qtype["gguf_iq2_xs"]
IQ1_S = ggml_tensor_qtype["gguf_iq1_s"]
IQ1_M = ggml_tensor_qtype["gguf_iq1_m"]
Q2_K = ggml_tensor_qtype["q2_k"]
Q4_K = ggml_tensor_qtype["q4_k"]
Q5_K = ggml_tensor_qtype["q5_k"]
Q6_K = ggml_tensor_qtype["q6_k"]
FP6_K = ggml_tensor_qtype["fp6_k"]
SYM_INT4_RTN = ggml_tensor_qtype["sym_int4_rtn"]
SYM_INT8_RTN = ggml_tensor_qtype["sym_int8_rtn"]
ASYM_INT4_RTN = ggml_tensor_qtype["asym_int4_rtn"]
WOQ_INT4 = ggml_tensor_qtype["woq_int4"]

RTN_QTYPES = (SYM_INT4_RTN, SYM_INT8_RTN, ASYM_INT4_RTN)
# formats quantized with the importance-matrix-aware entry point
K_QTYPES = (IQ2_XXS, IQ2_XS, Q2_K, IQ1_S, IQ1_M, Q4_K, Q6_K, Q5_K, FP6_K)
# formats with a tiled device layout and a fused MFMA kernel on gfx950
# (q4_k / q5_k are tiled as asym_int4 / asym_int5 with per-32 scale = d * sc6 and min = -dmin * m6)
T16_QTYPES = (SYM_INT4, ASYM_INT4, FP4, NF4, WOQ_INT4, SYM_INT5, ASYM_INT5, FP6, SYM_INT8, FP8E4, FP8E5, Q4_K, Q5_K)
# tiled formats whose device form cannot be turned back into the host blocks (the host copy is kept instead)
T16_ONE_WAY = (Q4_K, Q5_K)
DENSE_QTYPES = (FP16, BF16)


def qtype_name(qtype: int) -> str:
    return _id_to_name[qtype]


def qtype_id(name) -> int:
    if isinstance(name, int):
        return name
    if name in ggml_tensor_qtype:
        return ggml_tensor_qtype[name]
    if name in gguf_mixed_qtype:
        return gguf_mixed_qtype[name]
    raise ValueError(
        f"Unknown load_in_low_bit value: {name}, expected one of "
        f"{sorted(set(ggml_tensor_qtype) | set(gguf_mixed_qtype))}")
