"""`ipex_llm` import name for bigdl_amd (the reference aliases bigdl.llm -> ipex_llm the same way:
python/llm/src/ipex_llm/__init__.py:40-47).  Every sub-module of bigdl_amd resolves under this name."""
import importlib
import importlib.abc
import importlib.util
import sys

import bigdl_amd

_ALIASES = ("ipex_llm", "bigdl.llm")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        for a in _ALIASES:
            if fullname == a or fullname.startswith(a + "."):
                real = "bigdl_amd" + fullname[len(a):]
                try:
                    if importlib.util.find_spec(real) is None:
                        return None
                except (ImportError, ValueError):
                    return None
                return importlib.util.spec_from_loader(fullname, self, origin=real)
        return None

    def create_module(self, spec):
        mod = importlib.import_module(spec.origin)
        return mod

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = bigdl_amd
sys.modules.setdefault("bigdl.llm", bigdl_amd)
