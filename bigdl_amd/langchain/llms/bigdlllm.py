"""LangChain LLMs over single-file native models (reference: python/llm/src/ipex_llm/langchain/llms/bigdlllm.py:
``BigdlNativeLLM`` + the per-family ``LlamaLLM`` / ``BloomLLM`` / ``GptneoxLLM`` / ``ChatGLMLLM`` / ``StarcoderLLM``)."""
from __future__ import annotations

from typing import Any, List, Optional

from .._base import LLMBase


class _BaseCausalLM(LLMBase):
    ggml_model: str = "Llama"
    client: Any = None
    model_path: str = ""
    n_ctx: int = 512
    n_threads: Optional[int] = -1
    max_tokens: Optional[int] = 256
    temperature: Optional[float] = 0.8
    top_p: Optional[float] = 0.95
    top_k: Optional[int] = 40
    repeat_penalty: Optional[float] = 1.1
    stop: Optional[List[str]] = None
    echo: Optional[bool] = False
    streaming: bool = False

    def __init__(self, **kwargs: Any):
        family = kwargs.pop("model_family", None)
        super().__init__(**kwargs)
        if getattr(self, "client", None) is None:
            from ...ggml import model as M
            name = {"llama": "Llama", "bloom": "Bloom", "gptneox": "Gptneox", "chatglm": "ChatGLM",
                    "starcoder": "Starcoder"}.get(family, type(self).ggml_model)
            object.__setattr__(self, "client", getattr(M, name)(model_path=self.model_path, n_ctx=self.n_ctx,
                                                                n_threads=self.n_threads or -1))

    @property
    def _llm_type(self) -> str:
        return "bigdl-llm"

    @property
    def _default_params(self):
        return dict(max_tokens=self.max_tokens, temperature=self.temperature, top_p=self.top_p, top_k=self.top_k,
                    repeat_penalty=self.repeat_penalty, stop=self.stop or [], echo=self.echo)

    def _call(self, prompt: str, stop: Optional[List[str]] = None, run_manager=None, **kwargs: Any) -> str:
        params = {**self._default_params, **kwargs}
        if stop:
            params["stop"] = list(stop)
        return self.client(prompt, **params)["choices"][0]["text"]

    def get_num_tokens(self, text: str) -> int:
        return len(self.client.tokenize(text))


class BigdlNativeLLM(_BaseCausalLM):
    """``BigdlNativeLLM(model_path=..., model_family="llama")``"""


class LlamaLLM(_BaseCausalLM):
    ggml_model = "Llama"


class BloomLLM(_BaseCausalLM):
    ggml_model = "Bloom"


class GptneoxLLM(_BaseCausalLM):
    ggml_model = "Gptneox"


class ChatGLMLLM(_BaseCausalLM):
    ggml_model = "ChatGLM"


class StarcoderLLM(_BaseCausalLM):
    ggml_model = "Starcoder"
