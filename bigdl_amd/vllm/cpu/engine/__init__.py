"""from ipex_llm.vllm.cpu.engine import IPEXLLMClass as LLM (reference python/llm/src/ipex_llm/vllm/cpu/engine/__init__.py)."""
from ...engine import run_mp_engine  # noqa: F401


def __getattr__(name):
    from ... import engine
    return getattr(engine, name)
