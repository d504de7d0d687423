"""bigdl_amd: MI355X-native low-bit LLM library (capabilities of IPEX-LLM, the content of the BigDL fork).

Public surface mirrors the reference package root (python/llm/src/ipex_llm/__init__.py:22-47):
``optimize_model``, ``llm_convert``, ``llm_patch`` and the ``transformers`` sub-package; the same code is
importable as ``ipex_llm`` and ``bigdl.llm``.
"""
import os as _os
import sys as _sys

__version__ = "0.1.0"

# BIGDL_CHECK_DUPLICATE_IMPORT (default on, reference ipex_importer.py:27, 88-110): importing the library after a
# different copy of it was already imported under one of its other names is reported instead of silently mixing two
# sets of patched classes
if _os.environ.get("BIGDL_CHECK_DUPLICATE_IMPORT", "true").lower() in ("true", "1"):
    for _alias in ("ipex_llm", "bigdl.llm"):
        _m = _sys.modules.get(_alias)
        if _m is not None and getattr(_m, "__name__", "bigdl_amd") not in ("bigdl_amd", _alias) \
                and getattr(_m, "__file__", None) and "bigdl_amd" not in str(_m.__file__) \
                and _os.path.basename(_os.path.dirname(str(_m.__file__))) not in ("ipex_llm", "bigdl"):
            raise ImportError(f"a different package is already imported as {_alias}: {_m.__file__}")


# BIGDL_XPU_ALIAS=1: scripts written for the reference's "xpu" device run as they are (utils/xpu_alias.py)
if _os.environ.get("BIGDL_XPU_ALIAS", "0").lower() in ("true", "1"):
    from .utils.xpu_alias import enable_xpu_alias as _enable_xpu_alias
    _enable_xpu_alias()


def __getattr__(name):
    # lazy: importing the package must stay cheap (no torch/transformers import for the qtype table)
    if name in ("enable_xpu_alias", "disable_xpu_alias"):
        from .utils import xpu_alias
        return getattr(xpu_alias, name)
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
