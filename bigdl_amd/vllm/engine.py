"""vLLM engine classes that quantize the model into this package's low-bit format while vLLM loads it.

Reference: python/llm/src/ipex_llm/vllm/xpu/engine/engine.py:40-246 (``IPEXLLMAsyncLLMEngine``, ``IPEXLLMClass``,
``IPEXLLMLLMEngine``, ``run_mp_engine``) and its CPU twin python/llm/src/ipex_llm/vllm/cpu/engine/engine.py.  Each
class has vLLM's own signature plus ``load_in_low_bit``; before vLLM builds its workers the model runners' ``load_model``
is rebound (``model_convert._ipex_llm_convert``) so that every worker -- in this process, or a Ray / multiprocessing
worker that imports the patched module -- converts its shard on the host and then moves it to its GPU.

vLLM is imported lazily: the classes are created on first attribute access, so importing this module never needs vLLM
(on ROCm: the ROCm build of vLLM).  Without vLLM (or with ``IPEX_LLM_NOT_USE_VLLM=1``) ``IPEXLLMClass`` is
``offline.LLM``: the same ``LLM(...).generate(prompts, SamplingParams)`` interface on this package's continuous batcher."""
from __future__ import annotations

from .model_convert import _ipex_llm_convert

_CACHE = {}


def _need_vllm():
    try:
        import vllm  # noqa: F401
    except ImportError as e:                      # pragma: no cover - depends on the environment
        raise ImportError("this entry point needs vLLM (ROCm build) to be installed") from e


def _build():
    _need_vllm()
    from vllm import LLM
    from vllm.engine.async_llm_engine import AsyncLLMEngine
    from vllm.engine.llm_engine import LLMEngine

    class IPEXLLMAsyncLLMEngine(AsyncLLMEngine):
        @classmethod
        def from_engine_args(cls, engine_args, *args, load_in_low_bit: str = "sym_int4", **kwargs):
            _ipex_llm_convert(load_in_low_bit)
            return super().from_engine_args(engine_args, *args, **kwargs)

    class IPEXLLMLLMEngine(LLMEngine):
        @classmethod
        def from_engine_args(cls, engine_args, *args, load_in_low_bit: str = "sym_int4", **kwargs):
            _ipex_llm_convert(load_in_low_bit)
            return super().from_engine_args(engine_args, *args, **kwargs)

    class IPEXLLMClass(LLM):
        def __init__(self, *args, load_in_low_bit: str = "sym_int4", **kwargs):
            _ipex_llm_convert(load_in_low_bit)
            kwargs.pop("device", None)            # the reference passes device="xpu"; here the platform is ROCm
            super().__init__(*args, **kwargs)

    out = {"IPEXLLMAsyncLLMEngine": IPEXLLMAsyncLLMEngine, "IPEXLLMLLMEngine": IPEXLLMLLMEngine,
           "IPEXLLMClass": IPEXLLMClass}
    try:                                          # V1 engine classes exist from vLLM 0.7 on
        from vllm.v1.engine.async_llm import AsyncLLM
        from vllm.v1.engine.llm_engine import LLMEngine as V1LLMEngine

        class IPEXLLMAsyncV1Engine(AsyncLLM):
            @classmethod
            def from_engine_args(cls, engine_args, *args, load_in_low_bit: str = "sym_int4", **kwargs):
                _ipex_llm_convert(load_in_low_bit)
                return super().from_engine_args(engine_args, *args, **kwargs)

        class IPEXLLMLLMV1Engine(V1LLMEngine):
            @classmethod
            def from_engine_args(cls, engine_args, *args, load_in_low_bit: str = "sym_int4", **kwargs):
                _ipex_llm_convert(load_in_low_bit)
                return super().from_engine_args(engine_args, *args, **kwargs)
        out.update(IPEXLLMAsyncV1Engine=IPEXLLMAsyncV1Engine, IPEXLLMLLMV1Engine=IPEXLLMLLMV1Engine)
    except ImportError:
        pass
    return out


def run_mp_engine(engine_args, usage_context, ipc_path, load_in_low_bit: str, engine_alive=None):
    """Entry of vLLM's multiprocessing front end: convert, then run vLLM's own engine loop."""
    _need_vllm()
    _ipex_llm_convert(load_in_low_bit)
    from vllm.engine.multiprocessing.engine import run_mp_engine as _run
    args = (engine_args, usage_context, ipc_path) + ((engine_alive,) if engine_alive is not None else ())
    return _run(*args)


def _have_vllm() -> bool:
    import importlib.util
    import os
    return os.environ.get("IPEX_LLM_NOT_USE_VLLM", "") not in ("1", "true", "True") and importlib.util.find_spec("vllm") is not None


def __getattr__(name):
    if name == "IPEXLLMClass" and not _have_vllm():
        # without vLLM the offline entry point runs on this package's own batching engine (same call interface)
        from .offline import LLM
        return LLM
    if name.startswith("IPEXLLM"):
        if not _CACHE:
            _CACHE.update(_build())
        if name in _CACHE:
            return _CACHE[name]
    raise AttributeError(name)


__all__ = ["IPEXLLMAsyncLLMEngine", "IPEXLLMLLMEngine", "IPEXLLMClass", "IPEXLLMAsyncV1Engine", "IPEXLLMLLMV1Engine",
           "run_mp_engine"]
