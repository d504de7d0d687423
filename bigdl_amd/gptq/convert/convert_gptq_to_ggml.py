"""GPTQ checkpoint -> single quantized file (reference: python/llm/src/ipex_llm/gptq/convert/convert_gptq_to_ggml.py)."""
import os

from ...convert_model import llm_convert


def convert_gptq2ggml(model_path: str, output_path: str, tokenizer_path: str = None):
    out_dir = os.path.dirname(output_path) or "."
    kw = {"tokenizer_path": tokenizer_path} if tokenizer_path else {}
    written = llm_convert(model_path, out_dir, "llama", "int4", "gptq", **kw)
    if os.path.abspath(written) != os.path.abspath(output_path):
        os.replace(written, output_path)
    return output_path
