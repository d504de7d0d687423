from .bigdlllm import BigdlNativeLLM, BloomLLM, ChatGLMLLM, GptneoxLLM, LlamaLLM, StarcoderLLM  # noqa: F401
from .transformersllm import TransformersLLM  # noqa: F401
from .transformerspipelinellm import TransformersPipelineLLM  # noqa: F401

__all__ = ["TransformersLLM", "TransformersPipelineLLM", "BigdlNativeLLM", "LlamaLLM", "BloomLLM", "GptneoxLLM",
           "ChatGLMLLM", "StarcoderLLM"]
