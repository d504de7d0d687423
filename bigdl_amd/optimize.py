"""Generic-PyTorch API: ``optimize_model`` / ``load_low_bit`` / ``low_memory_init``.

Capability map to the reference python/llm/src/ipex_llm/optimize.py: ``_save_low_bit`` :41-57,
``low_memory_init`` :124-134, ``load_low_bit`` :137-194, ``optimize_model`` :197-265.
On-disk format (kept): ``bigdl_config.json`` = {"bigdl_transformers_low_bit": name} + ``pytorch_model.bin``.
"""
from __future__ import annotations

import json
import os
import types
from contextlib import contextmanager

import torch
from torch import nn

from . import qtypes as Q
from .utils.common import invalidInputError

CONFIG_NAME = "bigdl_config.json"
PYTORCH_MODEL_NAME = "pytorch_model.bin"


def _save_low_bit(self, save_dir, *args, **kwargs):
    invalidInputError(getattr(self, "_bigdl_config", {}).get("bigdl_transformers_low_bit", False),
                      "Detected this model is not a low-bit model, please use optimize_model first.")
    os.makedirs(save_dir, exist_ok=True)
    if isinstance(self, __import__("transformers").PreTrainedModel):
        from .transformers.model import save_low_bit as hf_save
        self.config.update({"bigdl_transformers_low_bit": self._bigdl_config["bigdl_transformers_low_bit"]})
        hf_save(self, save_dir, *args, **kwargs)
    else:
        device = next(self.parameters()).device
        self.to("cpu")
        sd = {k: (v.data if isinstance(v, nn.Parameter) else v) for k, v in self.state_dict().items()}
        torch.save(sd, os.path.join(save_dir, PYTORCH_MODEL_NAME))
        self.to(device)
    from .transformers.low_bit_linear import tensor_qtype_overrides
    with open(os.path.join(save_dir, CONFIG_NAME), "w") as f:
        json.dump(dict(self._bigdl_config, tensor_qtypes=tensor_qtype_overrides(self)), f)


@contextmanager
def low_memory_init():
    """Build models without allocating storage for weights that ``load_low_bit`` is about to replace by quantized
    ones.  The reference disables tensor allocation and lazy-unpickles the checkpoint (optimize.py:63-86, 124-134);
    here modules are constructed on the meta device, and HuggingFace's ``from_pretrained`` (which refuses to run under a
    meta-device context, and whose weight loading would not match the low-bit buffers anyway) returns the architecture
    built from the directory's config for the duration of the context:

        with low_memory_init():
            model = AutoModelForCausalLM.from_pretrained(saved_dir, torch_dtype="auto")
        model = load_low_bit(model, saved_dir)
    """
    import transformers
    base = transformers.PreTrainedModel
    original = base.__dict__["from_pretrained"]

    def meta_from_pretrained(cls, pretrained_model_name_or_path, *args, **kwargs):
        config = kwargs.pop("config", None)
        if config is None:
            config = transformers.AutoConfig.from_pretrained(pretrained_model_name_or_path,
                                                             trust_remote_code=kwargs.get("trust_remote_code", False))
        dtype = kwargs.get("torch_dtype", kwargs.get("dtype"))
        if dtype == "auto":
            dtype = getattr(config, "torch_dtype", None) or getattr(config, "dtype", None)
        with torch.device("meta"):
            model = cls(config)
        if isinstance(dtype, torch.dtype):
            model = model.to(dtype)
        return model.eval()

    base.from_pretrained = classmethod(meta_from_pretrained)
    try:
        with torch.device("meta"):
            yield
    finally:
        base.from_pretrained = original


def load_low_bit(model, model_path):
    """Fill an (empty or fp) ``model`` with the low-bit weights saved by ``model.save_low_bit(model_path)``."""
    from .transformers.convert import ggml_convert_low_bit
    from .transformers.model import _assign_state_dict, _materialize_meta_buffers
    from .transformers.utils import extract_local_archive_file, get_local_shard_files, load_state_dict
    invalidInputError(isinstance(model, nn.Module), "model should be a instance of `torch.nn.Module`")
    invalidInputError(os.path.isdir(model_path), "model_path should be an existing directory")
    cfg_path = os.path.join(model_path, CONFIG_NAME)
    invalidInputError(os.path.isfile(cfg_path), f"{CONFIG_NAME} not found in {model_path}")
    with open(cfg_path) as f:
        _config = json.load(f)
    low_bit = _config.get("bigdl_transformers_low_bit", None)
    invalidInputError(low_bit, "Detect this model is not a low-bit model, Please use `optimize_model` to get a "
                               "low-bit model, and serialize the model using save_low_bit first.")
    invalidInputError(low_bit in Q.ggml_tensor_qtype or low_bit in Q.gguf_mixed_qtype,
                      f"Unknown bigdl_transformers_low_bit value: {low_bit}")
    qtype = Q.qtype_id(low_bit)
    on_meta = any(p.device.type == "meta" for p in model.parameters())
    model = ggml_convert_low_bit(model, qtype=qtype, optimize_model=_config.get("optimize_llm", True),
                                 convert_shape_only=True, device="meta" if on_meta else "cpu",
                                 modules_to_not_convert=_config.get("modules_to_not_convert"))
    f, sharded = extract_local_archive_file(model_path)
    files = get_local_shard_files(model_path, f)[0] if sharded else [f]
    for ck in files:
        _assign_state_dict(model, load_state_dict(ck))
    _materialize_meta_buffers(model)
    from .transformers.low_bit_linear import apply_tensor_qtype_overrides
    apply_tensor_qtype_overrides(model, _config.get("tensor_qtypes"))
    if hasattr(model, "tie_weights"):
        try:
            head = model.get_output_embeddings()
        except Exception:
            head = None
        if head is None or type(head) is nn.Linear:          # (a low-bit head owns its packed blocks: never re-tied)
            try:
                model.tie_weights()
            except Exception:
                pass
    left = [n for n, p in model.named_parameters() if p.device.type == "meta"]
    if left == ["lm_head.weight"] and hasattr(model, "get_input_embeddings"):
        model.lm_head.weight = model.get_input_embeddings().weight
        left = []
    invalidInputError(not left, f"parameters missing from the checkpoint: {left[:5]}")
    model._bigdl_config = _config
    model.save_low_bit = types.MethodType(_save_low_bit, model)
    return model.eval()


def optimize_model(model, low_bit="sym_int4", optimize_llm=True, modules_to_not_convert=None, cpu_embedding=False,
                   **kwargs):
    """Quantize every linear of an arbitrary ``nn.Module`` and apply the LLM forward optimizations."""
    from .transformers.convert import ggml_convert_low_bit
    from .models.sd import is_diffusion_pipeline, optimize_diffusion_pipeline, unwrap_sentence_transformer
    if is_diffusion_pipeline(model):
        # diffusers pipelines are not modules: fused attention processor on the denoiser, bf16 guard on the SDXL VAE
        # (reference transformers/convert.py:1251-1263; its examples call optimize_model(pipe, low_bit=None))
        return optimize_diffusion_pipeline(model, low_bit=low_bit, modules_to_not_convert=modules_to_not_convert)
    inner = unwrap_sentence_transformer(model) if isinstance(model, nn.Module) else None
    if inner is not None:
        # a SentenceTransformer: its encoder is converted in place, pooling / normalisation modules stay
        # (reference convert.py:1265-1281)
        optimize_model(inner, low_bit=low_bit, optimize_llm=optimize_llm, modules_to_not_convert=modules_to_not_convert,
                       cpu_embedding=cpu_embedding, **kwargs)
        return model
    invalidInputError(isinstance(model, nn.Module), "model should be an instance of `torch.nn.Module`, "
                                                    f"but got {type(model)} at last.")
    if low_bit is None:
        low_bit = "bf16" if any(p.dtype == torch.bfloat16 for p in model.parameters()) else "fp16"
    invalidInputError(low_bit in Q.ggml_tensor_qtype or low_bit in Q.gguf_mixed_qtype,
                      f"Unknown load_in_low_bit value: {low_bit}, expected: sym_int4, asym_int4, sym_int5, asym_int5 "
                      f"or sym_int8 ...")
    devs = {p.device.type for p in model.parameters()}
    invalidInputError(devs <= {"cpu"},
                      "Expect model on device `cpu`, but got device type " f"{devs}")
    if kwargs.pop("replace_embedding", False):
        cpu_embedding = True
    torch_dtype = kwargs.get("torch_dtype", None)
    qtype = Q.qtype_id(low_bit)
    if torch_dtype is None:
        dts = {p.dtype for p in model.parameters() if p.is_floating_point()}
        torch_dtype = "auto" if len(dts) != 1 else next(iter(dts))
        if qtype == Q.FP16:
            torch_dtype = torch.float16
        elif qtype == Q.BF16:
            torch_dtype = torch.bfloat16
    model = ggml_convert_low_bit(model, qtype=qtype, optimize_model=optimize_llm,
                                 modules_to_not_convert=modules_to_not_convert, cpu_embedding=cpu_embedding,
                                 torch_dtype=torch_dtype)
    model._bigdl_config = {"bigdl_transformers_low_bit": low_bit, "optimize_llm": optimize_llm,
                           "modules_to_not_convert": modules_to_not_convert}
    if hasattr(model, "config") and hasattr(model.config, "update"):
        model.config.update({"bigdl_transformers_low_bit": low_bit})
    model.save_low_bit = types.MethodType(_save_low_bit, model)
    return model
