"""``llm_convert`` / ``llm-convert``: turn a HuggingFace checkpoint (dense or GPTQ) into one quantized model file.

The reference's legacy path produced per-family ggml ``.bin`` files through external ``quantize-<family>`` binaries
and ran them with ctypes llama.cpp runtimes (python/llm/src/ipex_llm/convert_model.py:31-144,
ggml/convert_model.py:26-103, utils/convert_util.py).  The equivalent single-file format here is GGUF written by this
package: ``int4`` -> q4_0 blocks, ``int8`` -> q8_0 blocks, produced in-process by the C++ quantizer; the file is loaded
back by ``BigdlNativeForCausalLM`` / the family wrappers in ``bigdl_amd.ggml.model`` and runs on the native engine.
"""
from __future__ import annotations

import argparse
import os
import time

import torch

from .utils.common import invalidInputError

_OUTTYPE = {"int4": "sym_int4", "int5": "sym_int5", "int8": "sym_int8", "q4_0": "sym_int4", "q4_1": "asym_int4",
            "q5_0": "sym_int5", "q5_1": "asym_int5", "q8_0": "sym_int8", "fp16": "fp16"}
_FAMILIES = ("llama", "bloom", "gptneox", "starcoder", "chatglm", "mistral", "mixtral", "qwen2")


def _gguf_arch(model_type: str) -> str:
    return {"llama": "llama", "mistral": "llama", "mixtral": "llama", "qwen2": "qwen2", "bloom": "bloom", "falcon": "falcon",
            "mpt": "mpt"}.get(model_type, model_type)


def _permute(w, n_head):
    rows = w.shape[0]
    return w.reshape(n_head, 2, rows // n_head // 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)


def export_gguf(model, tokenizer, path: str, low_bit: str = "sym_int4"):
    """Write a (dense) llama-layout / bloom HF model as GGUF with block-quantized matrices."""
    from . import qtypes as Q
    from .transformers.gguf import gguf as G
    cfg = model.config
    arch = _gguf_arch(cfg.model_type)
    invalidInputError(arch in ("llama", "qwen2"), f"export of model_type {cfg.model_type} to a single file is not supported")
    wtype = G.GGML_F16 if low_bit in ("fp16", "bf16") else G.QTYPE_TO_GGML[Q.qtype_id(low_bit)]
    ftype = {G.GGML_F16: 1, G.GGML_Q4_0: 2, G.GGML_Q4_1: 3, G.GGML_Q8_0: 7, G.GGML_Q5_0: 8, G.GGML_Q5_1: 9,
             G.GGML_Q2_K: 10, G.GGML_Q4_K: 15, G.GGML_Q5_K: 17, G.GGML_Q6_K: 18}[wtype]
    wr = G.GGUFWriter()
    wr.add("general.architecture", arch)
    wr.add("general.name", getattr(cfg, "_name_or_path", "") or cfg.model_type)
    wr.add("general.file_type", ftype)
    p = arch + "."
    for k, v in (("context_length", cfg.max_position_embeddings), ("embedding_length", cfg.hidden_size),
                 ("block_count", cfg.num_hidden_layers), ("feed_forward_length", cfg.intermediate_size),
                 ("attention.head_count", cfg.num_attention_heads),
                 ("attention.head_count_kv", getattr(cfg, "num_key_value_heads", cfg.num_attention_heads))):
        wr.add(p + k, int(v))
    wr.add(p + "attention.layer_norm_rms_epsilon", float(cfg.rms_norm_eps))
    rp = getattr(cfg, "rope_parameters", None) or {}
    wr.add(p + "rope.freq_base", float(rp.get("rope_theta", getattr(cfg, "rope_theta", 10000.0))))
    experts = getattr(cfg, "num_local_experts", 0) if cfg.model_type == "mixtral" else 0
    if experts:
        wr.add(p + "expert_count", int(experts))
        wr.add(p + "expert_used_count", int(cfg.num_experts_per_tok))
    if tokenizer is not None:
        vocab = tokenizer.get_vocab()
        toks = [None] * max(len(vocab), cfg.vocab_size)
        for t, i in vocab.items():
            if i < len(toks):
                toks[i] = t
        toks = [t if t is not None else f"<pad_{i}>" for i, t in enumerate(toks)]
        is_spm = any("▁" in t for t in toks[:2000])
        wr.add("tokenizer.ggml.model", "llama" if is_spm else "gpt2")
        wr.add("tokenizer.ggml.tokens", toks)
        wr.add("tokenizer.ggml.scores", [0.0] * len(toks))
        special = set(tokenizer.all_special_ids)
        wr.add("tokenizer.ggml.token_type", [3 if i in special else 1 for i in range(len(toks))])
        merges = None
        try:
            import json
            merges = json.loads(tokenizer.backend_tokenizer.to_str())["model"].get("merges")
        except Exception:
            pass
        if merges:
            wr.add("tokenizer.ggml.merges", [m if isinstance(m, str) else " ".join(m) for m in merges])
        for key, attr in (("bos_token_id", "bos_token_id"), ("eos_token_id", "eos_token_id"),
                          ("unknown_token_id", "unk_token_id"), ("padding_token_id", "pad_token_id")):
            v = getattr(tokenizer, attr, None)
            if v is not None:
                wr.add("tokenizer.ggml." + key, int(v))
    sd = model.state_dict()
    unperm = arch == "llama"
    n_head, n_kv = cfg.num_attention_heads, getattr(cfg, "num_key_value_heads", cfg.num_attention_heads)

    def mat(name, t, permute_heads=0):
        t = t.float()
        if permute_heads:
            t = _permute(t, permute_heads)
        k = t.shape[-1]
        wt = wtype
        if wt not in (G.GGML_F16, G.GGML_F32) and k % (256 if wt in (G.GGML_Q2_K, G.GGML_Q4_K, G.GGML_Q5_K, G.GGML_Q6_K) else 32):
            wt = G.GGML_F16
        wr.add_tensor(name, t, wt)

    mat("token_embd.weight", sd["model.embed_tokens.weight"])
    wr.add_tensor("output_norm.weight", sd["model.norm.weight"].float(), G.GGML_F32)
    mat("output.weight", sd.get("lm_head.weight", sd["model.embed_tokens.weight"]))
    for i in range(cfg.num_hidden_layers):
        q = f"model.layers.{i}."
        wr.add_tensor(f"blk.{i}.attn_norm.weight", sd[q + "input_layernorm.weight"].float(), G.GGML_F32)
        wr.add_tensor(f"blk.{i}.ffn_norm.weight", sd[q + "post_attention_layernorm.weight"].float(), G.GGML_F32)
        mat(f"blk.{i}.attn_q.weight", sd[q + "self_attn.q_proj.weight"], n_head if unperm else 0)
        mat(f"blk.{i}.attn_k.weight", sd[q + "self_attn.k_proj.weight"], n_kv if unperm else 0)
        mat(f"blk.{i}.attn_v.weight", sd[q + "self_attn.v_proj.weight"])
        mat(f"blk.{i}.attn_output.weight", sd[q + "self_attn.o_proj.weight"])
        for b in ("q", "k", "v"):
            key = q + f"self_attn.{b}_proj.bias"
            if key in sd:
                wr.add_tensor(f"blk.{i}.attn_{b}.bias", sd[key].float(), G.GGML_F32)
        if experts:
            wr.add_tensor(f"blk.{i}.ffn_gate_inp.weight", sd[q + "mlp.gate.weight"].float(), G.GGML_F32)
            gu = sd[q + "mlp.experts.gate_up_proj"]
            inter = gu.shape[1] // 2
            mat(f"blk.{i}.ffn_gate_exps.weight", gu[:, :inter].contiguous())
            mat(f"blk.{i}.ffn_up_exps.weight", gu[:, inter:].contiguous())
            mat(f"blk.{i}.ffn_down_exps.weight", sd[q + "mlp.experts.down_proj"])
        else:
            for a, b in (("ffn_gate", "mlp.gate_proj"), ("ffn_up", "mlp.up_proj"), ("ffn_down", "mlp.down_proj")):
                mat(f"blk.{i}.{a}.weight", sd[q + b + ".weight"])
    wr.write(path)
    return path


def quantize_checkpoint_file(input_path: str, output_path: str, low_bit: str = "sym_int4"):
    """Re-quantize a GGUF file (any readable tensor types) to ``low_bit`` blocks."""
    from . import qtypes as Q
    from .transformers.gguf import gguf as G
    ld = G.GGUFFileLoader(input_path)
    wtype = G.QTYPE_TO_GGML[Q.qtype_id(low_bit)]
    wr = G.GGUFWriter()
    for k, v in ld.config.items():
        if k != "general.alignment":
            wr.add(k, v)
    for name in ld.tensor_loader.names():
        t = ld.tensor_loader.load(name, torch.float32)
        is_matrix = t.dim() >= 2 and name.endswith(".weight") and "norm" not in name and t.shape[-1] % 32 == 0 \
            and "ffn_gate_inp" not in name
        wr.add_tensor(name, t, wtype if is_matrix else G.GGML_F32)
    wr.write(output_path)
    return output_path


def llm_convert(model, outfile, model_family, outtype="int4", model_format="pth", **kwargs):
    """Convert a HuggingFace checkpoint directory (``model_format='pth'``) or a GPTQ checkpoint directory
    (``model_format='gptq'``, llama + int4 only) into one quantized file under ``outfile`` (a directory).
    Returns the path of the written file."""
    invalidInputError(model_family in _FAMILIES, f"model_family must be one of {_FAMILIES}")
    invalidInputError(outtype in _OUTTYPE, f"outtype must be one of {sorted(_OUTTYPE)}")
    invalidInputError(model_format in ("pth", "gptq"), "model_format must be 'pth' or 'gptq'")
    invalidInputError(os.path.isdir(model), f"{model} is not a checkpoint directory")
    os.makedirs(outfile, exist_ok=True)
    from transformers import AutoTokenizer
    if model_format == "gptq":
        invalidInputError(model_family == "llama" and outtype == "int4",
                          "Convert GPTQ models should always specify `--model-family llama --dtype int4`.")
        from .transformers import AutoModelForCausalLM
        from .transformers.low_bit_linear import LowBitLinear, dequant_weight
        qm = AutoModelForCausalLM.from_pretrained(model, load_in_low_bit="asym_int4", optimize_model=False)
        for name, mod in list(qm.named_modules()):                       # back to dense for the writer
            if isinstance(mod, LowBitLinear):
                parent = qm.get_submodule(name.rpartition(".")[0]) if "." in name else qm
                dense = torch.nn.Linear(mod.in_len, mod.out_len, bias=mod.bias is not None)
                dense.weight = torch.nn.Parameter(dequant_weight(mod.weight, torch.float32), requires_grad=False)
                if mod.bias is not None:
                    dense.bias = torch.nn.Parameter(mod.bias.data.float(), requires_grad=False)
                setattr(parent, name.rpartition(".")[2], dense)
        hf, low_bit = qm, "asym_int4"
        fname = f"bigdl_llm_{model_family}_{outtype.lower()}_from_gptq.gguf"
        tok_path = kwargs.get("tokenizer_path") or model
    else:
        from transformers import AutoModelForCausalLM as HFAuto
        hf = HFAuto.from_pretrained(model, dtype=torch.float32)
        low_bit = _OUTTYPE[outtype]
        fname = f"bigdl_llm_{model_family}_{_OUTTYPE[outtype]}.gguf"
        tok_path = model
    try:
        tok = AutoTokenizer.from_pretrained(tok_path)
    except Exception:
        tok = None
    t0 = time.time()
    out = export_gguf(hf, tok, os.path.join(outfile, fname), low_bit)
    print(f"Converted {model} -> {out} in {time.time() - t0:.1f}s")
    return out


def main():
    ap = argparse.ArgumentParser(description="Model Convert Parameters")
    ap.add_argument("model", help="path to a HuggingFace (or GPTQ) checkpoint directory")
    ap.add_argument("-o", "--outfile", required=True, help="output directory")
    ap.add_argument("-x", "--model-family", required=True, choices=_FAMILIES)
    ap.add_argument("-f", "--model-format", default="pth", choices=["pth", "gptq"])
    ap.add_argument("-t", "--outtype", default="int4", choices=sorted(_OUTTYPE))
    ap.add_argument("-p", "--tmp-path", default=None)
    ap.add_argument("-k", "--tokenizer-path", default=None)
    a = ap.parse_args()
    kw = {k: v for k, v in dict(tmp_path=a.tmp_path, tokenizer_path=a.tokenizer_path).items() if v}
    llm_convert(a.model, a.outfile, a.model_family, a.outtype, a.model_format, **kw)


if __name__ == "__main__":
    main()
