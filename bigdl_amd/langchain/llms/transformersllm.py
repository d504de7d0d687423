"""``TransformersLLM``: a LangChain LLM over a low-bit ``AutoModelForCausalLM``
(reference: python/llm/src/ipex_llm/langchain/llms/transformersllm.py:64-270)."""
from __future__ import annotations

from typing import Any, List, Optional

from .._base import LLMBase, enforce_stop_tokens

DEFAULT_MODEL_ID = "gpt2"


class TransformersLLM(LLMBase):
    model_id: str = DEFAULT_MODEL_ID
    model_kwargs: Optional[dict] = None
    model: Any = None
    tokenizer: Any = None
    streaming: bool = True

    @classmethod
    def from_model_id(cls, model_id: str, model_kwargs: Optional[dict] = None, device_map: str = "cpu",
                      tokenizer_id: str = None, **kwargs: Any):
        """Load with ``load_in_4bit=True`` (or the ``load_in_low_bit`` given in model_kwargs)."""
        from transformers import AutoTokenizer
        from ...transformers import AutoModel, AutoModelForCausalLM
        mk = dict(model_kwargs or {})
        tok = AutoTokenizer.from_pretrained(tokenizer_id or model_id, **{k: v for k, v in mk.items() if k == "trust_remote_code"})
        low = {"load_in_low_bit": mk.pop("load_in_low_bit")} if "load_in_low_bit" in mk else {"load_in_4bit": True}
        try:
            model = AutoModelForCausalLM.from_pretrained(model_id, **low, **mk)
        except Exception:
            model = AutoModel.from_pretrained(model_id, **low, **mk)
        if device_map not in ("cpu", None):
            model = model.half().to(device_map)
        mk.pop("trust_remote_code", None)
        return cls(model_id=model_id, model=model, tokenizer=tok, model_kwargs=mk, **kwargs)

    @classmethod
    def from_model_id_low_bit(cls, model_id: str, model_kwargs: Optional[dict] = None, device_map: str = "cpu",
                              tokenizer_id: str = None, **kwargs: Any):
        """Load a directory written by ``save_low_bit``."""
        from transformers import AutoTokenizer
        from ...transformers import AutoModelForCausalLM
        mk = dict(model_kwargs or {})
        tok = AutoTokenizer.from_pretrained(tokenizer_id or model_id)
        model = AutoModelForCausalLM.load_low_bit(model_id, **mk)
        if device_map not in ("cpu", None):
            model = model.half().to(device_map)
        return cls(model_id=model_id, model=model, tokenizer=tok, model_kwargs=mk, **kwargs)

    @property
    def _identifying_params(self):
        return {"model_id": self.model_id, "model_kwargs": self.model_kwargs}

    @property
    def _llm_type(self) -> str:
        return "BigDL-llm"

    def _call(self, prompt: str, stop: Optional[List[str]] = None, run_manager=None, **kwargs: Any) -> str:
        import torch
        ids = self.tokenizer(prompt, return_tensors="pt").input_ids
        ids = ids.to(next(p for p in self.model.parameters() if p.numel() > 0).device)
        kwargs.setdefault("max_new_tokens", 64)
        with torch.inference_mode():
            out = self.model.generate(ids, **kwargs)
        text = self.tokenizer.decode(out[0, ids.shape[1]:], skip_special_tokens=True)
        if stop:
            text = enforce_stop_tokens(text, stop)
        return text
