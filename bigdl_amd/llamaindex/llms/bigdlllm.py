"""LlamaIndex ``CustomLLM`` over a low-bit model (reference: python/llm/src/ipex_llm/llamaindex/llms/bigdlllm.py:90
``IpexLLM`` with ``from_model_id`` / ``from_model_id_low_bit``, ``complete`` / ``stream_complete`` / ``chat``).
Derives from ``llama_index.core.llms.CustomLLM`` when that package is importable."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Iterator, Optional, Sequence

try:
    from llama_index.core.llms import CompletionResponse, CustomLLM, LLMMetadata
    from llama_index.core.llms.callbacks import llm_completion_callback
    HAVE_LLAMA_INDEX = True
except Exception:
    HAVE_LLAMA_INDEX = False

    @dataclass
    class CompletionResponse:
        text: str
        delta: Optional[str] = None
        raw: Any = None

    @dataclass
    class LLMMetadata:
        context_window: int = 3900
        num_output: int = 256
        model_name: str = "unknown"
        is_chat_model: bool = False

    class CustomLLM:
        def __init__(self, **kw):
            for k, v in kw.items():
                object.__setattr__(self, k, v)

    def llm_completion_callback():
        return lambda f: f

DEFAULT_CONTEXT_WINDOW, DEFAULT_NUM_OUTPUTS = 3900, 256


class IpexLLM(CustomLLM):
    model_name: str = "unknown"
    tokenizer_name: str = "unknown"
    context_window: int = DEFAULT_CONTEXT_WINDOW
    max_new_tokens: int = DEFAULT_NUM_OUTPUTS
    device_map: str = "cpu"
    generate_kwargs: Optional[dict] = None
    completion_to_prompt: Optional[Callable[[str], str]] = None
    messages_to_prompt: Optional[Callable[[Sequence[Any]], str]] = None

    def __init__(self, model_name: str, tokenizer_name: Optional[str] = None, context_window: int = DEFAULT_CONTEXT_WINDOW,
                 max_new_tokens: int = DEFAULT_NUM_OUTPUTS, load_low_bit: bool = False, low_bit: str = "sym_int4",
                 device_map: str = "cpu", model: Any = None, tokenizer: Any = None, generate_kwargs: Optional[dict] = None,
                 completion_to_prompt=None, messages_to_prompt=None, model_kwargs: Optional[dict] = None, **kw):
        from ...transformers import AutoModelForCausalLM
        if model is None:
            mk = dict(model_kwargs or {})
            if load_low_bit:
                model = AutoModelForCausalLM.load_low_bit(model_name, **mk)
            else:
                model = AutoModelForCausalLM.from_pretrained(model_name, load_in_low_bit=low_bit, **mk)
            if device_map not in ("cpu", None):
                model = model.half().to(device_map)
        if tokenizer is None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer_name or model_name)
        fields = dict(model_name=model_name, tokenizer_name=tokenizer_name or model_name, context_window=context_window,
                      max_new_tokens=max_new_tokens, device_map=device_map, generate_kwargs=generate_kwargs or {},
                      completion_to_prompt=completion_to_prompt, messages_to_prompt=messages_to_prompt)
        if HAVE_LLAMA_INDEX:
            super().__init__(**{k: v for k, v in fields.items() if k in type(self).model_fields})
        else:
            super().__init__(**fields)
        object.__setattr__(self, "_model", model)
        object.__setattr__(self, "_tokenizer", tokenizer)

    @classmethod
    def from_model_id(cls, model_name: str, **kw):
        return cls(model_name=model_name, load_low_bit=False, **kw)

    @classmethod
    def from_model_id_low_bit(cls, model_name: str, **kw):
        return cls(model_name=model_name, load_low_bit=True, **kw)

    @classmethod
    def class_name(cls) -> str:
        return "IpexLLM"

    @property
    def metadata(self):
        return LLMMetadata(context_window=self.context_window, num_output=self.max_new_tokens, model_name=self.model_name)

    def _generate_text(self, prompt: str, **kwargs):
        import torch
        if self.completion_to_prompt is not None and not kwargs.pop("formatted", False):
            prompt = self.completion_to_prompt(prompt)
        kwargs.pop("formatted", None)
        ids = self._tokenizer(prompt, return_tensors="pt").input_ids
        ids = ids.to(next(p for p in self._model.parameters() if p.numel() > 0).device)
        gk = {"max_new_tokens": self.max_new_tokens, **(self.generate_kwargs or {}), **kwargs}
        with torch.inference_mode():
            out = self._model.generate(ids, **gk)
        return self._tokenizer.decode(out[0, ids.shape[1]:], skip_special_tokens=True)

    @llm_completion_callback()
    def complete(self, prompt: str, formatted: bool = False, **kwargs: Any) -> CompletionResponse:
        return CompletionResponse(text=self._generate_text(prompt, formatted=formatted, **kwargs))

    @llm_completion_callback()
    def stream_complete(self, prompt: str, formatted: bool = False, **kwargs: Any) -> Iterator[CompletionResponse]:
        from threading import Thread
        from transformers import TextIteratorStreamer
        streamer = TextIteratorStreamer(self._tokenizer, skip_prompt=True, skip_special_tokens=True)
        t = Thread(target=self._generate_text, args=(prompt,), kwargs=dict(formatted=formatted, streamer=streamer, **kwargs))
        t.start()
        text = ""
        for piece in streamer:
            text += piece
            yield CompletionResponse(text=text, delta=piece)
        t.join()

    def chat(self, messages, **kwargs: Any):
        prompt = self.messages_to_prompt(messages) if self.messages_to_prompt is not None else \
            "\n".join(f"{getattr(m, 'role', 'user')}: {getattr(m, 'content', m)}" for m in messages) + "\nassistant: "
        return self.complete(prompt, formatted=True, **kwargs)
