"""A minimal "model as a step function" interface shared by the decoding strategies (prompt lookup,
self-speculative, pipeline-parallel): prefill / forward n tokens / crop the KV cache.

Two implementations: ``NativeStepper`` drives the MI355X engine (hipGraph decode steps of q_len tokens, KV crop =
one counter write, bigdl_amd/models/decoder.py); ``HFStepper`` drives any HuggingFace causal LM through its own
forward with a croppable cache (CPU, or architectures without a native engine).  The reference reaches the same
ends by calling ``self(input_ids=..., past_key_values=...)`` and editing cache tensors in place
(python/llm/src/ipex_llm/transformers/speculative.py:478-546 ``_crop_past_key_values``).
"""
from __future__ import annotations

import torch


class HFStepper:
    def __init__(self, model, cache_factory=None):
        self.model = model
        self.cache = None
        self.cache_factory = cache_factory
        self.length = 0
        self.device = next(model.parameters()).device

    def _new_cache(self):
        if self.cache_factory is not None:
            return self.cache_factory()
        from .kv import DynamicNormalCache
        return DynamicNormalCache()

    def prefill(self, input_ids):
        """-> logits [B, V] of the last position"""
        self.cache = self._new_cache()
        self.length = 0
        return self.forward(input_ids)[:, -1, :]

    def forward(self, tokens):
        """tokens [B, n] appended to the cache -> logits [B, n, V] (fp32)"""
        tokens = tokens.to(self.device)
        n = tokens.shape[1]
        pos = torch.arange(self.length, self.length + n, device=self.device)[None].expand(tokens.shape[0], -1)
        kw = {}
        lm_head = getattr(self.model, "lm_head", None)
        saved = None
        if lm_head is not None and getattr(lm_head, "optimize_lm_head", False) and n > 10:
            saved, lm_head.optimize_lm_head = True, False   # verification needs every position's logits
        try:
            out = self.model(input_ids=tokens, past_key_values=self.cache, use_cache=True, position_ids=pos, **kw)
        finally:
            if saved:
                lm_head.optimize_lm_head = True
        self.cache = out.past_key_values
        self.length += n
        return out.logits.float()

    def crop(self, length: int):
        if length >= self.length:
            return
        for layer in self.cache.layers:
            if hasattr(layer, "_len"):
                layer.crop(max_length=length)
            else:
                layer.crop(self.length - length) if _crop_takes_count(layer) else layer.crop(length)
        self.length = length


def _crop_takes_count(layer) -> bool:
    import inspect
    try:
        return "tokens_to_remove" in inspect.signature(layer.crop).parameters
    except (TypeError, ValueError):
        return False


class NativeStepper:
    def __init__(self, engine, kv_fp8=False, capacity=None):
        self.engine = engine
        self.kv_fp8 = kv_fp8
        self.capacity = capacity
        self.device = engine.device

    @property
    def length(self):
        return self.engine.cache.host_len

    def prefill(self, input_ids, reserve: int = 512):
        input_ids = input_ids.to(self.device)
        b, s = input_ids.shape
        self.engine.ensure_cache(b, self.capacity or (s + reserve), fp8=self.kv_fp8)
        self.engine.cache.reset()
        return self.engine.prefill(input_ids)[:, -1, :]

    def forward(self, tokens):
        tokens = tokens.to(self.device)
        b, n = tokens.shape
        logits, _ = self.engine.decode(tokens.contiguous(), n)
        return logits.view(b, n, -1).clone()

    def crop(self, length: int):
        if length < self.length:
            self.engine.cache.crop(length)


def make_stepper(model, kv_fp8=None, capacity=None):
    """Native engine when the model has one and lives on the GPU, else the HF forward."""
    from ..models.hf_patch import get_native_engine, native_engine_supported
    from .utils import engine_kv_mode
    dev = next(model.parameters()).device
    if dev.type == "cuda" and native_engine_supported(model):
        return NativeStepper(get_native_engine(model), engine_kv_mode() if kv_fp8 is None else kv_fp8, capacity)
    return HFStepper(model)
