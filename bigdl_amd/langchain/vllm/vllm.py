"""``VLLM``: a LangChain LLM over the batched offline engine
(reference: python/llm/src/ipex_llm/langchain/vllm/vllm.py:40-229: a LangChain ``BaseLLM`` whose client is
``IPEXLLMClass`` and whose ``_generate`` hands ALL prompts of a call to one ``client.generate``).  The client here is
``bigdl_amd.vllm.engine.IPEXLLMClass`` too: vLLM's ``LLM`` when vLLM is installed, this package's continuous-batching
``offline.LLM`` otherwise; either way a ``batch`` of prompts is one batched generation, not a loop."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

from .._base import LLMBase


class VLLM(LLMBase):
    model: str = ""
    tensor_parallel_size: int = 1
    trust_remote_code: bool = False
    n: int = 1
    top_k: int = -1
    top_p: float = 1.0
    temperature: float = 1.0
    max_new_tokens: int = 512
    stop: Optional[List[str]] = None
    ignore_eos: bool = False
    dtype: str = "auto"
    load_in_low_bit: str = "sym_int4"
    device: str = "xpu"
    vllm_kwargs: Optional[Dict[str, Any]] = None
    client: Any = None

    def __init__(self, **kwargs: Any):
        super().__init__(**kwargs)
        if self.client is None:
            from ...vllm.engine import __getattr__ as _engine_attr
            llm_cls = _engine_attr("IPEXLLMClass")
            client = llm_cls(model=self.model, tensor_parallel_size=self.tensor_parallel_size,
                             trust_remote_code=self.trust_remote_code, dtype=self.dtype,
                             load_in_low_bit=self.load_in_low_bit, device=self.device, **(self.vllm_kwargs or {}))
            object.__setattr__(self, "client", client)

    @property
    def _default_params(self) -> Dict[str, Any]:
        return {"n": self.n, "max_tokens": self.max_new_tokens, "top_k": self.top_k, "top_p": self.top_p,
                "temperature": self.temperature, "stop": self.stop, "ignore_eos": self.ignore_eos}

    @property
    def _llm_type(self) -> str:
        return "vllm"

    def _sampling_params(self, stop, kwargs):
        try:
            from vllm import SamplingParams
        except ImportError:
            from ...vllm.offline import SamplingParams
        params = {**self._default_params, **kwargs}
        if stop:
            params["stop"] = stop
        return SamplingParams(**params)

    def generate_texts(self, prompts: List[str], stop: Optional[List[str]] = None, **kwargs: Any) -> List[str]:
        """All prompts in one batched generation."""
        outs = self.client.generate(prompts, self._sampling_params(stop, kwargs))
        return [o.outputs[0].text for o in outs]

    def _call(self, prompt: str, stop: Optional[List[str]] = None, run_manager=None, **kwargs: Any) -> str:
        return self.generate_texts([prompt], stop=stop, **kwargs)[0]

    def batch(self, prompts: List[str], config=None, **kwargs: Any) -> List[str]:
        return self.generate_texts(list(prompts), **kwargs)

    def _generate(self, prompts: List[str], stop: Optional[List[str]] = None, run_manager=None, **kwargs: Any):
        """LangChain's batched hook: one ``LLMResult`` with a generation list per prompt."""
        from langchain_core.outputs import Generation, LLMResult
        return LLMResult(generations=[[Generation(text=t)] for t in self.generate_texts(prompts, stop=stop, **kwargs)])
