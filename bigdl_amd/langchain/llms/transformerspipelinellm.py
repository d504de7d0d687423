"""``TransformersPipelineLLM``: LangChain LLM over a HF ``pipeline`` built on a low-bit model
(reference: python/llm/src/ipex_llm/langchain/llms/transformerspipelinellm.py)."""
from __future__ import annotations

from typing import Any, List, Optional

from .._base import LLMBase, enforce_stop_tokens

VALID_TASKS = ("text2text-generation", "text-generation", "summarization")


class TransformersPipelineLLM(LLMBase):
    pipeline: Any = None
    model_id: str = "gpt2"
    model_kwargs: Optional[dict] = None
    pipeline_kwargs: Optional[dict] = None

    @classmethod
    def from_model_id(cls, model_id: str, task: str, model_kwargs: Optional[dict] = None,
                      pipeline_kwargs: Optional[dict] = None, **kwargs: Any):
        from transformers import AutoTokenizer
        from transformers import pipeline as hf_pipeline
        from ...transformers import AutoModelForCausalLM, AutoModelForSeq2SeqLM
        if task not in VALID_TASKS:
            raise ValueError(f"Got invalid task {task}, currently only {VALID_TASKS} are supported")
        mk = dict(model_kwargs or {})
        tok = AutoTokenizer.from_pretrained(model_id)
        auto = AutoModelForCausalLM if task == "text-generation" else AutoModelForSeq2SeqLM
        model = auto.from_pretrained(model_id, load_in_4bit=True, **mk)
        pipe = hf_pipeline(task=task, model=model, tokenizer=tok, **(pipeline_kwargs or {}))
        return cls(pipeline=pipe, model_id=model_id, model_kwargs=mk, pipeline_kwargs=pipeline_kwargs, **kwargs)

    @property
    def _llm_type(self) -> str:
        return "BigDL-llm"

    def _call(self, prompt: str, stop: Optional[List[str]] = None, run_manager=None, **kwargs: Any) -> str:
        response = self.pipeline(prompt)
        task = self.pipeline.task
        if task == "text-generation":
            text = response[0]["generated_text"][len(prompt):]
        elif task == "text2text-generation":
            text = response[0]["generated_text"]
        else:
            text = response[0]["summary_text"]
        return enforce_stop_tokens(text, stop) if stop else text
