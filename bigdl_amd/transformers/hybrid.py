"""CPU-GPU hybrid placement: chosen submodules (typically the routed experts of a very large MoE) stay in host memory
as low-bit weights and run on the CPU, the rest of the model (attention, dense MLPs, shared experts, lm_head) runs on
the GPU; activations hop over per layer.

Reference: python/llm/example/GPU/CPU-GPU-Hybrid-DeepSeek-R1-Inference/generate.py:36-48, 108-114, 208-216, 250-251
(attention and the first dense MLPs on the GPU in fp16, the int4 experts on the CPU).  A 288 GB MI355X holds a 4-bit
671B model's hot part but not all of it; the same split applies.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch
from torch import nn


class HostResident(nn.Module):
    """Wraps a module that must stay on the CPU: ``.to(device)`` / ``.cuda()`` of the parent leave it alone, and its
    forward moves tensor arguments to the host and the result back to the caller's device (pinned staging buffers, so
    the copies are asynchronous DMA over the host link)."""

    def __init__(self, module: nn.Module, compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.inner = module.to("cpu")
        self.compute_dtype = compute_dtype
        self.hops = 0

    def _apply(self, fn, recurse=True):          # parameters never leave the host
        return self

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("inner"), name)

    @staticmethod
    def _map(obj, fn):
        if isinstance(obj, torch.Tensor):
            return fn(obj)
        if isinstance(obj, (list, tuple)):
            return type(obj)(HostResident._map(o, fn) for o in obj)
        if isinstance(obj, dict):
            return {k: HostResident._map(v, fn) for k, v in obj.items()}
        return obj

    def forward(self, *args, **kwargs):
        dev, dt = None, None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_floating_point():
                dev, dt = a.device, a.dtype
                break

        def down(t):
            t = t.detach().to("cpu")
            return t.to(self.compute_dtype) if t.is_floating_point() else t

        def up(t):
            return t.to(dev, dt if t.is_floating_point() else None, non_blocking=True)

        out = self.inner(*self._map(args, down), **self._map(kwargs, down))
        self.hops += 1
        return self._map(out, up) if dev is not None else out


def _default_match(name: str, module: nn.Module) -> bool:
    from .low_bit_experts import LowBitExperts
    if isinstance(module, LowBitExperts):
        return True
    leaf = name.rsplit(".", 1)[-1]
    return leaf == "experts" and isinstance(module, (nn.ModuleList, nn.Module))


def offload_to_cpu(model: nn.Module, match: Optional[Callable[[str, nn.Module], bool]] = None,
                   skip_layers: Iterable[int] = (), compute_dtype: torch.dtype = torch.float32) -> int:
    """Wrap every submodule selected by ``match(name, module)`` (default: the routed-expert containers of MoE blocks)
    in ``HostResident``.  Call it on the converted (low-bit) model before ``model.to("cuda")``.  ``skip_layers``:
    decoder layer indices whose experts should go to the GPU with the rest.  Returns the number of wrapped modules."""
    match = match or _default_match
    skip = {f".{i}." for i in skip_layers}
    n = 0
    for name, module in list(model.named_modules()):
        if isinstance(module, HostResident) or not name or not match(name, module):
            continue
        if any(s in f".{name}." for s in skip):
            continue
        parent_name, _, leaf = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        if isinstance(parent, HostResident) or any(isinstance(m, HostResident) for m in _ancestors(model, parent_name)):
            continue
        setattr(parent, leaf, HostResident(module, compute_dtype))
        n += 1
    return n


def _ancestors(model, name):
    parts = name.split(".") if name else []
    for i in range(1, len(parts) + 1):
        yield model.get_submodule(".".join(parts[:i]))
