"""Legacy import name: ``bigdl.llm`` is the same package as ``ipex_llm`` / ``bigdl_amd``
(reference alias: python/llm/src/ipex_llm/__init__.py:40-47)."""
import sys

import ipex_llm  # noqa: F401  (installs the alias finder for sub-modules)
import bigdl_amd as llm  # noqa: F401

sys.modules[__name__ + ".llm"] = llm
