"""LoRA / QLoRA / QA-LoRA / ReLoRA fine-tuning on frozen low-bit base weights.

Capability map to the reference python/llm/src/ipex_llm/transformers/qlora.py: ``LoraLowBitLinear`` :66-143
(incl. the QA-LoRA average pooling of the adapter input), ``LoraBF16Linear`` :146-211, ``LoraConfig`` with
``training_mode`` :253-263, ``TrainingArguments`` :269-279, ``get_peft_model`` :282-301,
``prepare_model_for_kbit_training`` :304-352, ``PeftModel.from_pretrained`` :355-369, ``cast_lora_weight`` :377-391.

The reference monkey-patches the ``peft`` package; this build owns the (small) LoRA machinery so that it does not
depend on it: adapter injection, adapter-only state dict in peft's file layout (``adapter_config.json`` +
``adapter_model.safetensors`` with ``base_model.model.<module>.lora_A.weight`` keys), merging.

Compute path on MI355X: the frozen base matmul is ``MatMulLowBit`` (forward: dequant to bf16 + hipBLASLt GEMM for
training-sized batches; backward: grad @ dequant(W), no weight gradient), adapters run in bf16.
"""
from __future__ import annotations

import json
import logging
import math
import os
import re
from dataclasses import asdict, dataclass, field
from typing import List, Optional, Union

import torch
from torch import nn

from .low_bit_linear import BF16Linear, FP16Linear, LowBitLinear, dequant_weight, get_qk_size
from .utils import get_autocast_dtype

LOG = logging.getLogger("bigdl_amd.qlora")


@dataclass
class LoraConfig:
    r: int = 8
    lora_alpha: int = 8
    lora_dropout: float = 0.0
    target_modules: Optional[Union[List[str], str]] = None
    bias: str = "none"
    task_type: Optional[str] = "CAUSAL_LM"
    fan_in_fan_out: bool = False
    modules_to_save: Optional[List[str]] = None
    init_lora_weights: bool = True
    use_rslora: bool = False
    inference_mode: bool = False
    base_model_name_or_path: Optional[str] = None
    peft_type: str = "LORA"
    training_mode: str = field(default="qlora", metadata={"help": "qlora | qalora | lora | relora"})

    def to_dict(self):
        d = asdict(self)
        if isinstance(d["target_modules"], (set, tuple)):
            d["target_modules"] = list(d["target_modules"])
        return d

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "adapter_config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2)

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "adapter_config.json")) as f:
            d = json.load(f)
        known = {k: v for k, v in d.items() if k in cls.__dataclass_fields__}
        return cls(**known)


class LoraLayer:
    """Adapter state shared by the LoRA linears."""

    def _init_lora(self, base_layer, adapter_name, r, lora_alpha, lora_dropout, init_lora_weights, use_rslora,
                   adapter_in_features):
        self.base_layer = base_layer
        self.r, self.lora_alpha, self.scaling = {}, {}, {}
        self.lora_dropout = nn.ModuleDict()
        self.lora_A = nn.ModuleDict()
        self.lora_B = nn.ModuleDict()
        self.merged = False
        self.disable_adapters = False
        self.active_adapters = [adapter_name]
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.update_layer(adapter_name, r, lora_alpha, lora_dropout, init_lora_weights, use_rslora, adapter_in_features)

    def update_layer(self, adapter_name, r, lora_alpha, lora_dropout, init_lora_weights=True, use_rslora=False,
                     adapter_in_features=None):
        if r <= 0:
            raise ValueError(f"`r` should be a positive integer value but the value passed is {r}")
        self.r[adapter_name], self.lora_alpha[adapter_name] = r, lora_alpha
        self.lora_dropout[adapter_name] = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()
        k = adapter_in_features or self.in_features
        self.lora_A[adapter_name] = nn.Linear(k, r, bias=False)
        self.lora_B[adapter_name] = nn.Linear(r, self.out_features, bias=False)
        self.scaling[adapter_name] = lora_alpha / math.sqrt(r) if use_rslora else lora_alpha / r
        if init_lora_weights:
            nn.init.kaiming_uniform_(self.lora_A[adapter_name].weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B[adapter_name].weight)
        dev = next((p.device for p in self.base_layer.parameters() if p.device.type != "meta"), None)
        if dev is not None:
            self.lora_A[adapter_name].to(dev)
            self.lora_B[adapter_name].to(dev)

    def _adapter_delta(self, x, pool, expected_dtype=None):
        out = None
        for name in self.active_adapters:
            if name not in self.lora_A:
                continue
            a, b = self.lora_A[name], self.lora_B[name]
            xa = x.to(a.weight.dtype)
            d = b(a(self.lora_dropout[name](pool(xa)))) * self.scaling[name]
            out = d if out is None else out + d
        if out is not None and expected_dtype is not None:
            out = out.to(expected_dtype)
        return out


class FusedLoraLowBit(torch.autograd.Function):
    """``y = x W^T + s * dropout(x) A^T B^T`` on a frozen low-bit ``W`` as ONE autograd node (GPU training path of
    ``LoraLowBitLinear``).  The module-by-module form costs about a dozen small launches per adapted linear and step
    (dropout, two adapter products, scale, add, their backward twins and the casts between fp32 adapters and bf16
    activations: 2,900 launches per step with seven targets on Llama-2-7B); here the scale and the add ride in the
    adapter products (``addmm_`` with ``alpha``), the adapters are cast once (8 x K values) and the activations never
    leave bf16.  Math of the reference's LoraLowBitLinear.forward (python/llm/src/ipex_llm/transformers/qlora.py:57-105)."""

    @staticmethod
    def forward(ctx, x2d, weight, out_len, a_w, b_w, scale, p, training):
        from .low_bit_linear import lowbit_matmul
        y = lowbit_matmul(x2d, weight, out_len, for_training=True)
        if p > 0.0 and training:
            xd, mask = torch.ops.aten.native_dropout(x2d, p, True)
        else:
            xd, mask = x2d, None
        a16, b16 = a_w.to(x2d.dtype), b_w.to(x2d.dtype)
        r = a16.shape[0]
        # rank-8 / rank-16 products on the one-pass kernels of csrc/hip/lora_ops.hip (BIGDL_LORA_KERNELS=0: library GEMMs)
        ctx.native = _LORA_KERNELS and _hip_lora_ok(xd, y, r)
        if ctx.native:
            from ..ops import hip
            t = hip.lora_down(xd, a16, r, alpha=scale)         # [M, r], the adapter scale folded in here
            hip.lora_up(t, b16, y)                             # y += t @ B^T
        else:
            t = (xd @ a16.t()) * scale                         # [M, r]
            y.addmm_(t, b16.t())
        ctx.save_for_backward(xd, t, mask, a16, b16)
        ctx.weight, ctx.scale, ctx.p = weight, scale, p
        ctx.adapter_dtypes = (a_w.dtype, b_w.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .low_bit_linear import dequant_weight
        xd, t, mask, a16, b16 = ctx.saved_tensors
        gy = gy.contiguous()
        native = ctx.native and gy.dtype == xd.dtype
        if native:
            from ..ops import hip
            gt = hip.lora_down(gy, b16, a16.shape[0], alpha=ctx.scale)                   # [M, r] = scale * gy @ B
        else:
            gt = (gy @ b16) * ctx.scale                        # [M, r]
        # (t carries the adapter scale; gradients leave in the adapters' own dtype: no fp32 round trip for bf16 adapters)
        g_b = (gy.t() @ t).to(ctx.adapter_dtypes[1]) if ctx.needs_input_grad[4] else None      # [N, r]
        g_a = (gt.t() @ xd).to(ctx.adapter_dtypes[0]) if ctx.needs_input_grad[3] else None    # [r, K]
        gx = None
        if ctx.needs_input_grad[0]:
            gx = gy @ dequant_weight(ctx.weight, gy.dtype)
            if native:
                # gx += (gt @ A) * mask / (1 - p): one pass over gx, the mask read as the bytes it is stored in
                hip.lora_up(gt, a16, gx, alpha=1.0 if mask is None else 1.0 / (1.0 - ctx.p), mask=mask)
            elif mask is None:
                gx.addmm_(gt, a16)
            else:
                gx.addcmul_(gt @ a16, mask.to(gy.dtype), value=1.0 / (1.0 - ctx.p))
        return gx, None, None, g_a, g_b, None, None, None


_LORA_KERNELS = os.environ.get("BIGDL_LORA_KERNELS", "1") != "0"


def _hip_lora_ok(x, y, rank) -> bool:
    try:
        from ..ops import hip
    except Exception:
        return False
    return hip.lora_ops_ok(x, rank) and hip.lora_ops_ok(y, rank) and x.dtype == y.dtype


class LoraLowBitLinear(nn.Module, LoraLayer):
    """LoRA on a frozen low-bit linear.  With ``qa_lora`` the adapter sees the input average-pooled over each
    quantization block (QA-LoRA), so the merged update stays representable per block."""

    def __init__(self, base_layer, adapter_name="default", r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 qa_lora: bool = False, fan_in_fan_out: bool = False, init_lora_weights=True, use_rslora=False, **kwargs):
        nn.Module.__init__(self)
        qk = get_qk_size(kwargs.get("qtype", getattr(base_layer, "qtype", None))) if qa_lora else 1
        self._init_lora(base_layer, adapter_name, r, lora_alpha, lora_dropout, init_lora_weights, use_rslora,
                        adapter_in_features=base_layer.in_features // qk)
        self.fan_in_fan_out = fan_in_fan_out
        self.qa_pool = nn.AvgPool1d(qk) if qa_lora else nn.Identity()

    def forward(self, x: torch.Tensor):
        autocast_dtype = get_autocast_dtype(x)
        if x.device.type == "cuda":
            x = x.to(torch.bfloat16)          # adapters and the dequantized base GEMM run in bf16 on the GPU
        elif autocast_dtype is not None:
            x = x.to(autocast_dtype)
        if self._fusable(x):
            name = self.active_adapters[0]
            base = self.base_layer
            drop = self.lora_dropout[name]
            x2d = x.contiguous().view(-1, x.shape[-1])
            y = FusedLoraLowBit.apply(x2d, base.weight, base.out_len, self.lora_A[name].weight, self.lora_B[name].weight,
                                      float(self.scaling[name]), float(getattr(drop, "p", 0.0)), self.training)
            y = y.view(x.shape[:-1] + (base.out_len,))
            return y if base.bias is None else y + base.bias.to(y.dtype)
        result = self.base_layer.forward(x)
        if self.disable_adapters or self.merged:
            return result
        delta = self._adapter_delta(x, self.qa_pool, result.dtype)
        return result if delta is None else result + delta

    def _fusable(self, x):
        """one adapter on a plain low-bit linear, on the GPU, in a step that takes gradients (``BIGDL_FUSED_LORA=0``: off)"""
        base = self.base_layer
        return (x.device.type == "cuda" and x.dtype in (torch.bfloat16, torch.float16) and torch.is_grad_enabled()
                and not (self.disable_adapters or self.merged) and len(self.active_adapters) == 1
                and self.active_adapters[0] in self.lora_A and isinstance(self.qa_pool, nn.Identity)
                and type(base).__name__ == "LowBitLinear" and getattr(base, "mp_group", None) is None
                and not getattr(base, "act_order", False) and getattr(base.weight, "quantized", False)
                and 0 not in x.shape and os.environ.get("BIGDL_FUSED_LORA", "1") != "0")


class LoraBF16Linear(nn.Module, LoraLayer):
    """LoRA on a bf16 / fp16 base linear (``training_mode="lora"``)."""

    def __init__(self, base_layer, adapter_name="default", r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 fan_in_fan_out: bool = False, init_lora_weights=True, use_rslora=False, **kwargs):
        nn.Module.__init__(self)
        self._init_lora(base_layer, adapter_name, r, lora_alpha, lora_dropout, init_lora_weights, use_rslora, None)
        self.fan_in_fan_out = fan_in_fan_out

    def forward(self, x: torch.Tensor):
        autocast_dtype = get_autocast_dtype(x)
        if x.device.type == "cuda":
            x = x.to(torch.bfloat16)
        elif autocast_dtype is not None:
            x = x.to(autocast_dtype)
        result = self.base_layer.forward(x)
        if self.disable_adapters or self.merged:
            return result
        delta = self._adapter_delta(x, nn.Identity(), result.dtype)
        return result if delta is None else result + delta


class LoraDenseLinear(LoraBF16Linear):
    """LoRA on a plain ``nn.Linear`` that was not converted (e.g. excluded modules)."""


def _matches(name: str, targets) -> bool:
    if isinstance(targets, str):
        return re.fullmatch(targets, name) is not None
    return any(name == t or name.endswith("." + t) for t in targets)


def _create_new_module(lora_config: LoraConfig, adapter_name, target, **kwargs):
    """Pick the LoRA wrapper for a base layer (reference qlora.py:214-247)."""
    common = dict(r=lora_config.r, lora_alpha=lora_config.lora_alpha, lora_dropout=lora_config.lora_dropout,
                  fan_in_fan_out=lora_config.fan_in_fan_out, init_lora_weights=lora_config.init_lora_weights,
                  use_rslora=lora_config.use_rslora)
    if isinstance(target, LowBitLinear):
        return LoraLowBitLinear(target, adapter_name, qa_lora=lora_config.training_mode == "qalora",
                                qtype=target.qtype, **common)
    if isinstance(target, (BF16Linear, FP16Linear)):
        return LoraBF16Linear(target, adapter_name, **common)
    if isinstance(target, nn.Linear):
        return LoraDenseLinear(target, adapter_name, **common)
    raise ValueError(f"Target module {target} is not supported. Currently, only `LowBitLinear`, `BF16Linear`, "
                     f"`FP16Linear` and `torch.nn.Linear` are supported.")


class PeftModelForCausalLM(nn.Module):
    """Base model + injected adapters.  Attribute access falls through to the base model, so ``generate``,
    ``config`` etc. keep working."""

    def __init__(self, model, peft_config: LoraConfig, adapter_name: str = "default"):
        super().__init__()
        self.base_model = model
        self.peft_config = {adapter_name: peft_config}
        self.active_adapter = adapter_name
        self._inject(peft_config, adapter_name)
        for n, p in self.base_model.named_parameters():
            p.requires_grad = ("lora_" in n) and not peft_config.inference_mode
        if peft_config.modules_to_save:
            for n, p in self.base_model.named_parameters():
                if any(m in n for m in peft_config.modules_to_save) and p.dtype.is_floating_point:
                    p.requires_grad = True

    def _inject(self, cfg: LoraConfig, adapter_name):
        targets = cfg.target_modules
        if targets is None:
            targets = ["q_proj", "k_proj", "v_proj", "qkv_proj"]
        found = False
        for name, module in list(self.base_model.named_modules()):
            if isinstance(module, (LoraLayer,)) or not isinstance(module, nn.Linear) or not _matches(name, targets):
                continue
            parent_name, _, child = name.rpartition(".")
            parent = self.base_model.get_submodule(parent_name) if parent_name else self.base_model
            setattr(parent, child, _create_new_module(cfg, adapter_name, module))
            found = True
        if not found:
            raise ValueError(f"Target modules {targets} not found in the base model.")

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model, name)

    def forward(self, *args, **kwargs):
        return self.base_model(*args, **kwargs)

    def generate(self, *args, **kwargs):
        return self.base_model.generate(*args, **kwargs)

    def get_base_model(self):
        return self.base_model

    def print_trainable_parameters(self):
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        total = sum(p.numel() for p in self.parameters())
        print(f"trainable params: {trainable:,d} || all params: {total:,d} || trainable%: {100 * trainable / max(1, total):.4f}")
        return trainable, total

    # ---- adapter state --------------------------------------------------------------------------
    def adapter_state_dict(self, adapter_name=None):
        adapter_name = adapter_name or self.active_adapter
        out = {}
        for n, p in self.base_model.state_dict().items():
            if "lora_" in n:
                out["base_model.model." + n.replace(f".{adapter_name}.", ".")] = p.detach().cpu()
        cfg = self.peft_config[adapter_name]
        if cfg.modules_to_save:
            for n, p in self.base_model.state_dict().items():
                if any(m in n for m in cfg.modules_to_save) and "lora_" not in n:
                    out["base_model.model." + n] = p.detach().cpu()
        return out

    def save_pretrained(self, save_directory, safe_serialization=True, **kwargs):
        os.makedirs(save_directory, exist_ok=True)
        sd = self.adapter_state_dict()
        if safe_serialization:
            from safetensors.torch import save_file
            save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(save_directory, "adapter_model.safetensors"))
        else:
            torch.save(sd, os.path.join(save_directory, "adapter_model.bin"))
        cfg = self.peft_config[self.active_adapter]
        cfg.base_model_name_or_path = cfg.base_model_name_or_path or getattr(self.base_model.config, "_name_or_path", None)
        cfg.save_pretrained(save_directory)

    def load_adapter_state_dict(self, sd, adapter_name=None):
        adapter_name = adapter_name or self.active_adapter
        own = dict(self.base_model.named_parameters())
        own.update(dict(self.base_model.named_buffers()))
        missing = []
        for k, v in sd.items():
            name = k[len("base_model.model."):] if k.startswith("base_model.model.") else k
            name = re.sub(r"\.(lora_[AB])\.weight$", rf".\1.{adapter_name}.weight", name)
            if name in own:
                with torch.no_grad():
                    own[name].copy_(v.to(own[name].dtype))
            else:
                missing.append(k)
        if missing:
            LOG.warning(f"adapter keys without a matching module: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        return missing

    def merge_and_unload(self):
        """Fold the adapters into the base weights.  Low-bit base layers are dequantized, updated and returned as
        dense ``nn.Linear`` (re-quantize with ``optimize_model`` afterwards if wanted)."""
        for name, module in list(self.base_model.named_modules()):
            if not isinstance(module, LoraLayer):
                continue
            base = module.base_layer
            if isinstance(base, LowBitLinear):
                w = dequant_weight(base.weight, torch.float32)
            else:
                w = base.weight.data.float()
            for ad in module.active_adapters:
                delta = (module.lora_B[ad].weight.float() @ module.lora_A[ad].weight.float()) * module.scaling[ad]
                pool = getattr(module, "qa_pool", None)
                if isinstance(pool, nn.AvgPool1d):
                    qk = pool.kernel_size[0] if isinstance(pool.kernel_size, tuple) else pool.kernel_size
                    delta = delta.repeat_interleave(qk, dim=1) / qk
                w = w + delta.to(w.device)
            new = nn.Linear(w.shape[1], w.shape[0], bias=base.bias is not None)
            dt = torch.bfloat16 if w.device.type == "cuda" else torch.float32
            new.weight = nn.Parameter(w.to(dt), requires_grad=False)
            if base.bias is not None:
                new.bias = nn.Parameter(base.bias.data.to(dt), requires_grad=False)
            parent_name, _, child = name.rpartition(".")
            parent = self.base_model.get_submodule(parent_name) if parent_name else self.base_model
            setattr(parent, child, new)
        return self.base_model

    def disable_adapter_layers(self, disabled=True):
        for m in self.base_model.modules():
            if isinstance(m, LoraLayer):
                m.disable_adapters = disabled

    def disable_adapter(self):
        """Context manager: run the frozen base model (PEFT's ``with model.disable_adapter():``; the implicit
        reference model of DPO)."""
        import contextlib

        @contextlib.contextmanager
        def _ctx():
            self.disable_adapter_layers(True)
            try:
                yield
            finally:
                self.disable_adapter_layers(False)
        return _ctx()

    def gradient_checkpointing_enable(self, *a, **k):
        return self.base_model.gradient_checkpointing_enable(*a, **k)

    @property
    def device(self):
        return next(p.device for p in self.base_model.parameters() if p.numel() > 0)


LoraModel = PeftModelForCausalLM


def get_peft_model(model, peft_config: LoraConfig, adapter_name: str = "default"):
    from ..llm_patching import bigdl_patched
    if bigdl_patched == "Train":
        from .model import patched_training_mode
        if patched_training_mode is not None:
            peft_config.training_mode = patched_training_mode
    peft_model = PeftModelForCausalLM(model, peft_config, adapter_name)
    if peft_model.device.type == "cuda":
        cast_lora_weight(peft_model, torch.bfloat16)
        if os.environ.get("BIGDL_FUSED_TRAIN_OPS", "1") != "0":
            # norms, silu(g) * u and the rotary embedding as one kernel forward and one backward each
            from .fused_train_ops import bind_fused_training_ops
            bind_fused_training_ops(model)
    return peft_model


def prepare_model_for_kbit_training(model, use_gradient_checkpointing=True):
    """Freeze the base model, keep norms etc. in fp32, make the input embeddings' output require grad and turn on
    activation checkpointing."""
    is_gptq_quantized = getattr(model, "quantization_method", None) == "gptq"
    for _, param in model.named_parameters():
        param.requires_grad = False
    if not is_gptq_quantized:
        for module in model.modules():
            if not list(module.children()) and not isinstance(module, (nn.Linear, nn.Embedding)):
                for param in module.parameters():
                    if param.dtype in (torch.float16, torch.bfloat16):
                        param.data = param.data.to(torch.float32)
    if use_gradient_checkpointing:
        if hasattr(model, "enable_input_require_grads"):
            model.enable_input_require_grads()
        else:
            model.get_input_embeddings().register_forward_hook(lambda m, i, o: o.requires_grad_(True))
        model.gradient_checkpointing_enable()
    from .low_bit_linear import auto_train_dense_cache
    auto_train_dense_cache(model)
    return model


class PeftModel:
    @staticmethod
    def from_pretrained(model, model_id, adapter_name: str = "default", is_trainable: bool = False, config=None, **kwargs):
        cfg = config or LoraConfig.from_pretrained(model_id)
        cfg.inference_mode = not is_trainable
        pm = PeftModelForCausalLM(model, cfg, adapter_name)
        st = os.path.join(model_id, "adapter_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(model_id, "adapter_model.bin"), map_location="cpu", weights_only=True)
        pm.load_adapter_state_dict(sd, adapter_name)
        if pm.device.type == "cuda":
            cast_lora_weight(pm, torch.bfloat16)
        return pm


def cast_lora_weight(model, dtype=torch.bfloat16):
    for name, module in model.named_modules():
        if isinstance(module, LowBitLinear):
            module.compute_dtype = dtype
        if isinstance(module, LoraLayer):
            module.lora_A.to(dtype)
            module.lora_B.to(dtype)
        if isinstance(module, (BF16Linear, FP16Linear)):
            module.to(dtype)
            module.compute_dtype = dtype
        if "norm" in name and not list(module.children()):
            module.to(torch.float32)
        if ("lm_head" in name or "embed_tokens" in name) and hasattr(module, "weight") \
                and isinstance(module.weight, torch.Tensor) and module.weight.dtype == torch.float32 \
                and not isinstance(module, LowBitLinear):
            module.to(dtype)


supported_optim = ["adamw_hf", "adamw_torch", "adamw_torch_fused", "adafactor", "sgd", "adagrad", "rmsprop"]


def _training_arguments_base():
    from transformers import TrainingArguments as Base
    return Base


class TrainingArguments(_training_arguments_base()):
    """HF ``TrainingArguments`` pinned to bf16 compute and an optimizer that exists on this stack."""

    def __init__(self, *args, **kwargs):
        kwargs["fp16"] = False
        kwargs["bf16"] = kwargs.get("bf16", torch.cuda.is_available())
        if str(kwargs.get("optim", "adamw_torch")) not in supported_optim:
            LOG.info(f"{kwargs.get('optim')} is not supported yet and adamw_torch optimizer is used.")
            kwargs["optim"] = "adamw_torch"
        super().__init__(*args, **kwargs)
