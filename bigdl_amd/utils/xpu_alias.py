"""``"xpu"`` as a device name.

Scripts written for the reference say ``model.to("xpu")``, ``torch.tensor(..., device="xpu")``,
``torch.xpu.synchronize()`` (every example under python/llm/example/GPU does; the benchmark harness
python/llm/dev/benchmark/all-in-one/run.py:465-569 as well).  The GPU here is ``"cuda"`` (ROCm's device type in
PyTorch), and PyTorch rejects ``"xpu"`` in a build without that backend.  ``enable_xpu_alias()`` makes such scripts run
unchanged:

* a ``TorchFunctionMode`` that stays active for the rest of the process rewrites ``"xpu"`` / ``"xpu:N"`` /
  ``torch.device("xpu", N)`` wherever a torch function or tensor method receives a device -- positional for
  ``Tensor.to`` / ``Module.to`` style calls, ``device=`` everywhere (factory functions included);
* ``nn.Module.to`` / ``Module.xpu`` and ``Tensor.xpu`` are wrapped the same way (``Module.to`` parses its arguments
  before any torch function sees them);
* ``torch.xpu``'s query and housekeeping functions answer for the target device.

Off by default: the mode adds a Python hop to every torch call of the process (about a microsecond each; the native
engine's decode loop does not go through torch functions, the HuggingFace module path does).  ``BIGDL_XPU_ALIAS=1`` in
the environment enables it when the package is imported.
"""
from __future__ import annotations

import functools
import re
from typing import Optional

import torch
from torch.overrides import TorchFunctionMode

_XPU = re.compile(r"^xpu(?::(\d+))?$")
_state = {"mode": None, "target": None, "saved": []}


def _map(x, target: str):
    if isinstance(x, str):
        m = _XPU.match(x)
        if m:
            return target if (m.group(1) is None or target == "cpu") else f"{target}:{m.group(1)}"
    elif isinstance(x, torch.device) and x.type == "xpu":
        return torch.device(target) if (x.index is None or target == "cpu") else torch.device(target, x.index)
    return x


class _AliasMode(TorchFunctionMode):
    def __init__(self, target: str):
        super().__init__()
        self.target = target

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if "device" in kwargs:
            kwargs = dict(kwargs, device=_map(kwargs["device"], self.target))
        if args and any(isinstance(a, (str, torch.device)) for a in args):
            args = tuple(_map(a, self.target) for a in args)
        return func(*args, **kwargs)


def _wrap_method(cls, name: str, target: str):
    orig = getattr(cls, name)

    @functools.wraps(orig)
    def method(self, *args, **kwargs):
        if "device" in kwargs:
            kwargs["device"] = _map(kwargs["device"], target)
        return orig(self, *(_map(a, target) for a in args), **kwargs)
    _state["saved"].append((cls, name, orig))
    setattr(cls, name, method)


def enable_xpu_alias(target: Optional[str] = None) -> str:
    """Make ``"xpu"`` mean ``target`` ("cuda" when a GPU is present, else "cpu") from now on.  Returns the target."""
    if _state["mode"] is not None:
        return _state["target"]
    target = target or ("cuda" if torch.cuda.is_available() else "cpu")
    mode = _AliasMode(target)
    mode.__enter__()
    _state.update(mode=mode, target=target)
    _wrap_method(torch.nn.Module, "to", target)

    def module_xpu(self, device=None):
        return self.to(target if device is None else _map(f"xpu:{device}" if isinstance(device, int) else device, target))

    def tensor_xpu(self, device=None, non_blocking=False, **kw):
        return self.to(target if device is None else _map(f"xpu:{device}" if isinstance(device, int) else device, target),
                       non_blocking=non_blocking)
    for cls, name, fn in ((torch.nn.Module, "xpu", module_xpu), (torch.Tensor, "xpu", tensor_xpu)):
        _state["saved"].append((cls, name, getattr(cls, name, None)))
        setattr(cls, name, fn)
    # torch.xpu's housekeeping functions, answered by the target device
    cu = torch.cuda
    on_gpu = target == "cuda"
    table = {
        "is_available": (lambda: True),
        "device_count": (lambda: cu.device_count() if on_gpu else 1),
        "current_device": (lambda: cu.current_device() if on_gpu else 0),
        "set_device": (lambda d: cu.set_device(_map(d, target) if not isinstance(d, int) else d) if on_gpu else None),
        "synchronize": (lambda device=None: cu.synchronize() if on_gpu else None),
        "empty_cache": (lambda: cu.empty_cache() if on_gpu else None),
        "memory_allocated": (lambda device=None: cu.memory_allocated() if on_gpu else 0),
        "max_memory_allocated": (lambda device=None: cu.max_memory_allocated() if on_gpu else 0),
        "memory_reserved": (lambda device=None: cu.memory_reserved() if on_gpu else 0),
        "max_memory_reserved": (lambda device=None: cu.max_memory_reserved() if on_gpu else 0),
        "reset_peak_memory_stats": (lambda device=None: cu.reset_peak_memory_stats() if on_gpu else None),
        "manual_seed": (lambda s: cu.manual_seed(s) if on_gpu else torch.manual_seed(s)),
        "manual_seed_all": (lambda s: cu.manual_seed_all(s) if on_gpu else torch.manual_seed(s)),
        "get_device_name": (lambda device=None: cu.get_device_name() if on_gpu else "cpu"),
    }
    for name, fn in table.items():
        _state["saved"].append((torch.xpu, name, getattr(torch.xpu, name, None)))
        setattr(torch.xpu, name, fn)
    return target


def disable_xpu_alias():
    if _state["mode"] is None:
        return
    while _state["saved"]:
        owner, name, old = _state["saved"].pop()
        if old is None:
            if name in vars(owner):
                delattr(owner, name)
        else:
            setattr(owner, name, old)
    _state["mode"].__exit__(None, None, None)
    _state.update(mode=None, target=None)


__all__ = ["enable_xpu_alias", "disable_xpu_alias"]
