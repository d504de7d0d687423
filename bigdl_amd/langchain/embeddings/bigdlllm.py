"""Embeddings from single-file native models (reference: python/llm/src/ipex_llm/langchain/embeddings/bigdlllm.py)."""
from __future__ import annotations

from typing import Any, List, Optional

from .._base import EmbeddingsBase


class BigdlNativeEmbeddings(EmbeddingsBase):
    ggml_model: str = "Llama"
    client: Any = None
    model_path: str = ""
    n_ctx: int = 512
    n_threads: Optional[int] = -1

    def __init__(self, **kwargs: Any):
        family = kwargs.pop("model_family", None)
        super().__init__(**kwargs)
        if getattr(self, "client", None) is None:
            from ...ggml import model as M
            name = {"llama": "Llama", "bloom": "Bloom", "gptneox": "Gptneox", "starcoder": "Starcoder"}.get(
                family, type(self).ggml_model)
            object.__setattr__(self, "client", getattr(M, name)(model_path=self.model_path, n_ctx=self.n_ctx))

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return [list(map(float, self.client.embed(t))) for t in texts]

    def embed_query(self, text: str) -> List[float]:
        return list(map(float, self.client.embed(text)))


class LlamaEmbeddings(BigdlNativeEmbeddings):
    ggml_model = "Llama"


class BloomEmbeddings(BigdlNativeEmbeddings):
    ggml_model = "Bloom"


class GptneoxEmbeddings(BigdlNativeEmbeddings):
    ggml_model = "Gptneox"


class StarcoderEmbeddings(BigdlNativeEmbeddings):
    ggml_model = "Starcoder"
