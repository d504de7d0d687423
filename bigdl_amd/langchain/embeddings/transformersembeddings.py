"""Embeddings from a low-bit transformer's hidden states (reference:
python/llm/src/ipex_llm/langchain/embeddings/transformersembeddings.py: mean of the last hidden state;
``TransformersBgeEmbeddings`` uses the CLS token and a query instruction)."""
from __future__ import annotations

from typing import Any, List, Optional

from .._base import EmbeddingsBase


class TransformersEmbeddings(EmbeddingsBase):
    model: Any = None
    tokenizer: Any = None
    model_id: str = "gpt2"
    model_kwargs: Optional[dict] = None
    encode_kwargs: Optional[dict] = None

    @classmethod
    def from_model_id(cls, model_id: str, model_kwargs: Optional[dict] = None, device_map: str = "cpu", **kwargs: Any):
        from transformers import AutoTokenizer
        from ...transformers import AutoModel
        mk = dict(model_kwargs or {})
        tok = AutoTokenizer.from_pretrained(model_id)
        model = AutoModel.from_pretrained(model_id, load_in_4bit=True, **mk)
        if device_map not in ("cpu", None):
            model = model.half().to(device_map)
        return cls(model_id=model_id, model=model, tokenizer=tok, model_kwargs=mk, **kwargs)

    def _hidden(self, text: str):
        import torch
        enc = self.tokenizer(text, return_tensors="pt", **(self.encode_kwargs or {}))
        dev = next(p for p in self.model.parameters() if p.numel() > 0).device
        with torch.inference_mode():
            out = self.model(input_ids=enc.input_ids.to(dev))
        h = getattr(out, "last_hidden_state", None)
        if h is None:
            h = out.hidden_states[-1] if getattr(out, "hidden_states", None) else out[0]
        return h[0].float()

    def embed(self, text: str, **kwargs):
        return self._hidden(text).mean(0).cpu().tolist()

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return [self.embed(t) for t in texts]

    def embed_query(self, text: str) -> List[float]:
        return self.embed(text.replace("\n", " "))


class TransformersBgeEmbeddings(TransformersEmbeddings):
    query_instruction: str = "Represent this question for searching relevant passages: "

    def embed(self, text: str, **kwargs):
        import torch
        v = self._hidden(text)[0]
        return torch.nn.functional.normalize(v, dim=0).cpu().tolist()

    def embed_query(self, text: str) -> List[float]:
        return self.embed(self.query_instruction + text.replace("\n", " "))
