"""Hook vLLM's model loading so that every worker converts its (tensor-parallel shard of the) model to a low-bit
format of this package.

Reference: python/llm/src/ipex_llm/vllm/xpu/model_convert.py:65-168 (``_ipex_llm_convert(load_in_low_bit)`` patches
``XPUModelRunner.load_model`` / ``GPUModelRunner.load_model`` / ``RayWorkerWrapper``; ``get_load_function`` loads on
the CPU, runs ``optimize_model`` and moves to the device; ``padding_mlp`` for vLLM's merged MLP) and
vllm/xpu/engine/engine.py:40-246 (engine subclasses calling the hook first).  The converter in
``bigdl_amd.transformers.convert.is_linear_module`` already understands vLLM's Column/Row/QKV/MergedColumn parallel
linears and takes the tensor-parallel group for the all-reduce from vLLM.
"""
from __future__ import annotations

import torch

_CONVERTED = False


def get_load_function(low_bit: str):
    def _ipex_llm_load_model(self) -> None:
        from vllm.model_executor.model_loader import get_model
        from ..optimize import optimize_model
        from ..transformers import convert as C
        C._USE_VLLM = True
        kw = {}
        for name in ("vllm_config", "model_config", "device_config", "load_config", "lora_config", "parallel_config",
                     "scheduler_config", "cache_config"):
            if hasattr(self, name):
                kw[name] = getattr(self, name)
        target = None
        if "device_config" in kw:                    # build on the host, quantize, then move
            target = kw["device_config"].device
            kw["device_config"].device = torch.device("cpu")
        try:
            self.model = get_model(**{k: v for k, v in kw.items() if k in get_model.__code__.co_varnames})
        finally:
            if target is not None:
                kw["device_config"].device = target
        modules = None
        arch = type(self.model).__name__
        if "Qwen2" in arch and low_bit in ("fp16", "bf16"):
            modules = ["lm_head"]
        optimize_model(self.model, low_bit=low_bit, torch_dtype=self.model_config.dtype, modules_to_not_convert=modules)
        self.model = self.model.to(device=target or "cuda", dtype=self.model_config.dtype)
    return _ipex_llm_load_model


def _ipex_llm_convert(load_in_low_bit: str):
    """Patch vLLM's model runners; call before the engine is created."""
    global _CONVERTED
    if _CONVERTED:
        return
    try:
        import vllm  # noqa: F401
    except ImportError as e:
        raise ImportError("vLLM is not installed; `bigdl_amd.vllm` only adds low-bit loading to an existing vLLM") from e
    patched = 0
    for mod_name, cls_name in (("vllm.worker.model_runner", "ModelRunner"), ("vllm.worker.model_runner", "GPUModelRunnerBase"),
                               ("vllm.v1.worker.gpu_model_runner", "GPUModelRunner")):
        try:
            mod = __import__(mod_name, fromlist=[cls_name])
            setattr(getattr(mod, cls_name), "load_model", get_load_function(load_in_low_bit))
            patched += 1
        except Exception:
            continue
    if patched == 0:
        raise RuntimeError("no known vLLM model runner class found to patch")
    _CONVERTED = True


class IPEXLLMClass:
    """``LLM(...)`` with ``load_in_low_bit`` (reference engine.py ``IPEXLLMClass``)."""

    def __new__(cls, *args, load_in_low_bit: str = "sym_int4", **kwargs):
        _ipex_llm_convert(load_in_low_bit)
        from vllm import LLM
        return LLM(*args, **kwargs)
