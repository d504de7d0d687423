"""qtype table under its reference import path (python/llm/src/ipex_llm/ggml/quantize.py:28-61) and the
legacy file quantizer ``quantize(input_path, output_path, model_family, dtype)`` (:87-142), which here runs the
in-process C++ quantizer over a ggml/gguf fp16 file instead of spawning a per-family binary."""
from ..qtypes import ggml_tensor_qtype, gguf_mixed_qtype  # noqa: F401

_llama_quantize_type = {"q4_0": 2, "q4_1": 3, "q5_0": 8, "q5_1": 9, "q8_0": 7}
_quantize_type = {"llama": _llama_quantize_type, "bloom": {"q4_0": 2, "q4_1": 3},
                  "gptneox": _llama_quantize_type, "starcoder": _llama_quantize_type}
_name_to_qtype = {"q4_0": "sym_int4", "q4_1": "asym_int4", "q5_0": "sym_int5", "q5_1": "asym_int5", "q8_0": "sym_int8"}


def quantize(input_path: str, output_path: str, model_family: str, dtype: str = "q4_0"):
    from ..utils.common import invalidInputError
    invalidInputError(model_family in _quantize_type, f"{model_family} is not in the supported list "
                                                      f"{list(_quantize_type)}")
    invalidInputError(dtype in _quantize_type[model_family], f"{dtype} is not a supported dtype for {model_family}")
    from ..convert_model import quantize_checkpoint_file
    return quantize_checkpoint_file(input_path, output_path, _name_to_qtype[dtype])
