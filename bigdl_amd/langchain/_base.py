from __future__ import annotations

from typing import Any, List, Optional

try:                                                    # real LangChain
    from langchain_core.embeddings import Embeddings as _Embeddings
    from langchain_core.language_models.llms import LLM as _LLM
    HAVE_LANGCHAIN = True
except Exception:                                       # stand-ins with the same surface
    HAVE_LANGCHAIN = False

    class _LLM:
        def __init__(self, **kwargs):
            for k, v in kwargs.items():
                object.__setattr__(self, k, v)

        def invoke(self, prompt: str, stop: Optional[List[str]] = None, **kwargs: Any) -> str:
            return self._call(prompt, stop=stop, **kwargs)

        __call__ = invoke

        def batch(self, prompts: List[str], **kwargs: Any) -> List[str]:
            return [self.invoke(p, **kwargs) for p in prompts]

    class _Embeddings:
        def __init__(self, **kwargs):
            for k, v in kwargs.items():
                object.__setattr__(self, k, v)

LLMBase, EmbeddingsBase = _LLM, _Embeddings


def enforce_stop_tokens(text: str, stop: List[str]) -> str:
    import re
    return re.split("|".join(re.escape(s) for s in stop), text, maxsplit=1)[0]
