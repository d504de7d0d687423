"""bigdl_amd: MI355X-native low-bit LLM library (capabilities of IPEX-LLM, the content of the BigDL fork).

Public surface mirrors the reference package root (python/llm/src/ipex_llm/__init__.py:22-47):
``optimize_model``, ``llm_convert``, ``llm_patch`` and the ``transformers`` sub-package; the same code is
importable as ``ipex_llm`` and ``bigdl.llm``.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy: importing the package must stay cheap (no torch/transformers import for the qtype table)
    if name == "optimize_model":
        from .optimize import optimize_model
        return optimize_model
    if name == "llm_convert":
        from .convert_model import llm_convert
        return llm_convert
    if name in ("llm_patch", "llm_unpatch"):
        from . import llm_patching
        return getattr(llm_patching, name)
    raise AttributeError(name)
