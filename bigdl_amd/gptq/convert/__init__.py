from .convert_gptq_to_ggml import convert_gptq2ggml  # noqa: F401
