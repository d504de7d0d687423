"""``convert_model(input_path, output_path, model_family, dtype)`` under the reference's import path
(python/llm/src/ipex_llm/ggml/convert_model.py:26-103)."""
from ..convert_model import llm_convert


def convert_model(input_path: str, output_path: str, model_family: str, dtype: str = "int4", tmp_path: str = None):
    return llm_convert(input_path, output_path, model_family, outtype=dtype, model_format="pth")
