"""vLLM integration (ROCm build of vLLM): quantize the model a vLLM worker has just loaded into this package's low-bit
format before it is moved to the GPU.  See ``model_convert``."""
from .offline import LLM, CompletionOutput, RequestOutput, SamplingParams  # noqa: E402,F401
