"""FastChat model adapters: stock ``fastchat.serve.model_worker`` (and everything else that goes through
``fastchat.model.model_adapter.load_model``) loads its models through this package.

Reference: python/llm/src/ipex_llm/serving/fastchat/bigdl_llm_model.py (``BigDLLLMAdapter`` matches paths containing
"bigdl" and loads with ``load_in_4bit=True``; ``BigDLLMLOWBITAdapter`` matches "bigdl-lowbit" and calls
``load_low_bit``; ``patch_fastchat`` registers both and rebinds ``BaseModelAdapter.load_model``,
``ChatGLMAdapter.load_model`` and ``model_adapter.load_model`` to loaders that use this package's Auto classes).

FastChat is imported when ``patch_fastchat`` runs; the adapter classes themselves only need its ``BaseModelAdapter``
interface (``match`` / ``load_model`` / ``get_default_conv_template``), for which a stand-in is used when FastChat is not
installed, so the loaders can be used -- and are tested -- on their own.
"""
from __future__ import annotations

from typing import Optional

try:
    from fastchat.model.model_adapter import BaseModelAdapter
    HAVE_FASTCHAT = True
except Exception:
    HAVE_FASTCHAT = False

    class BaseModelAdapter:
        use_fast_tokenizer = True

        def match(self, model_path: str) -> bool:
            return True

        def get_default_conv_template(self, model_path: str):
            return None

is_fastchat_patched = False
_originals = []          # (owner, attribute, original) of everything patch_fastchat rebinds


def _tokenizer(model_path: str, **kw):
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(model_path, **kw)


def _placed(model, device: Optional[str]):
    """FastChat moves a model with ``model.to(device)`` after loading; low-bit weights want fp16 activations on the GPU."""
    import torch
    if device in ("cuda", "xpu", "gpu") and torch.cuda.is_available():
        return model.half().to("cuda")
    return model


def load_model_base(self, model_path: str, from_pretrained_kwargs: dict):
    """Replacement of ``BaseModelAdapter.load_model``: any causal LM, converted while it loads."""
    kw = dict(from_pretrained_kwargs)
    revision = kw.get("revision", "main")
    tokenizer = _tokenizer(model_path, use_fast=getattr(self, "use_fast_tokenizer", True), revision=revision)
    from ...transformers import AutoModelForCausalLM
    if not any(k in kw for k in ("load_in_4bit", "load_in_low_bit")):
        kw["load_in_4bit"] = True
    return AutoModelForCausalLM.from_pretrained(model_path, low_cpu_mem_usage=True, **kw), tokenizer


def load_model_chatglm(self, model_path: str, from_pretrained_kwargs: dict):
    """Replacement of ``ChatGLMAdapter.load_model`` (remote-code model class: ``AutoModel``)."""
    kw = dict(from_pretrained_kwargs)
    tokenizer = _tokenizer(model_path, trust_remote_code=True, revision=kw.get("revision", "main"))
    from ...transformers import AutoModel
    kw.setdefault("load_in_4bit", True)
    return AutoModel.from_pretrained(model_path, trust_remote_code=True, **kw), tokenizer


def load_model(model_path: str, device: str = "cuda", num_gpus: int = 1, max_gpu_memory: Optional[str] = None,
               load_8bit: bool = False, cpu_offloading: bool = False, gptq_config=None, awq_config=None,
               revision: str = "main", debug: bool = False, **ignored):
    """Replacement of ``fastchat.model.model_adapter.load_model``: picks the adapter the usual way, loads through it and
    places the result (one GPU: the whole model; the 8-bit / offloading / multi-GPU device-map options of the original
    have no meaning for a model that is already low-bit and are ignored)."""
    if HAVE_FASTCHAT:
        from fastchat.model.model_adapter import get_model_adapter
        adapter = get_model_adapter(model_path)
    else:
        adapter = next(a for a in (BigDLLMLOWBITAdapter(), BigDLLLMAdapter(), _AnyModelAdapter()) if a.match(model_path))
    import torch
    kwargs = {"torch_dtype": torch.float16 if device in ("cuda", "xpu", "gpu") else "auto", "revision": revision}
    model, tokenizer = adapter.load_model(model_path, kwargs)
    model = _placed(model, device)
    if debug:
        print(model)
    return model, tokenizer


class BigDLLLMAdapter(BaseModelAdapter):
    """Model adapter for checkpoints to be converted on load (path contains "bigdl")."""

    def match(self, model_path: str):
        return "bigdl" in model_path

    def load_model(self, model_path: str, from_pretrained_kwargs: dict):
        kw = dict(from_pretrained_kwargs)
        tokenizer = _tokenizer(model_path, use_fast=False, revision=kw.get("revision", "main"), trust_remote_code=True)
        from ...transformers import AutoModelForCausalLM
        model = AutoModelForCausalLM.from_pretrained(model_path, load_in_4bit=True, low_cpu_mem_usage=True, **kw)
        return model, tokenizer


class BigDLLMLOWBITAdapter(BaseModelAdapter):
    """Model adapter for directories written by ``save_low_bit`` (path contains "bigdl-lowbit")."""

    def match(self, model_path: str):
        return "bigdl-lowbit" in model_path

    def load_model(self, model_path: str, from_pretrained_kwargs: dict):
        tokenizer = _tokenizer(model_path, use_fast=False, revision=from_pretrained_kwargs.get("revision", "main"))
        from ...transformers import AutoModelForCausalLM
        return AutoModelForCausalLM.load_low_bit(model_path), tokenizer


class _AnyModelAdapter(BaseModelAdapter):
    load_model = load_model_base


def patch_fastchat():
    """Register the two adapters (the low-bit one first: its pattern contains the other's) and rebind FastChat's loaders."""
    global is_fastchat_patched
    if is_fastchat_patched:
        return
    from fastchat.model import model_adapter
    from fastchat.model.model_adapter import ChatGLMAdapter, register_model_adapter
    register_model_adapter(BigDLLMLOWBITAdapter)
    register_model_adapter(BigDLLLMAdapter)
    # registered adapters are tried in order and the catch-all ones come first in FastChat's own list
    adapters = getattr(model_adapter, "model_adapters", None)
    if adapters is not None:
        ours = [a for a in adapters if isinstance(a, (BigDLLMLOWBITAdapter, BigDLLLMAdapter))]
        rest = [a for a in adapters if a not in ours]
        adapters[:] = ours + rest
    for owner, name, new in ((BaseModelAdapter, "load_model", load_model_base),
                             (ChatGLMAdapter, "load_model", load_model_chatglm), (model_adapter, "load_model", load_model)):
        _originals.append((owner, name, getattr(owner, name, None)))
        setattr(owner, name, new)
    is_fastchat_patched = True


def unpatch_fastchat():
    global is_fastchat_patched
    while _originals:
        owner, name, old = _originals.pop()
        if old is not None:
            setattr(owner, name, old)
        elif name in vars(owner):
            delattr(owner, name)
    is_fastchat_patched = False


__all__ = ["BigDLLLMAdapter", "BigDLLMLOWBITAdapter", "load_model", "load_model_base", "load_model_chatglm",
           "patch_fastchat", "unpatch_fastchat"]
