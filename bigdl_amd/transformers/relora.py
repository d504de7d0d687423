"""ReLoRA: train LoRA adapters, every ``relora_steps`` merge them into the (low-bit) base weights, re-initialise the
adapters, prune the optimizer state and restart the learning-rate warm-up.

Capability map to the reference python/llm/src/ipex_llm/transformers/relora.py: ``ReLoRATrainer`` :64-108,
``reset_optimizer`` :128-146, ``ReLoRACallback`` :149-283, ``ReLoRAScheduler`` :286-316, ``lora_delta_weight``
:334-346, ``find_lora_modules`` :349-363, ``update_weights`` / ``merge_and_save`` :366-452.

The reference merges by rewriting safetensors shards of the fp16 checkpoint on disk and re-quantizing from them.
Here the merge happens in memory on the GPU: dequantize the block-quantized weight, add the delta, re-quantize with
the host library and re-tile -- the full-precision checkpoint is never needed again after the first load.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch
from torch.optim.lr_scheduler import LRScheduler
from transformers import Trainer, TrainerCallback

from .low_bit_linear import FP4Params, LowBitLinear, dequant_weight
from .qlora import LoraLayer


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def is_main_process():
    return not is_distributed() or torch.distributed.get_rank() == 0


def reset_optimizer(optimizer: torch.optim.Optimizer, reset_params=None, prune_ratio: float = 0.9):
    """Magnitude-prune the Adam moments of the (re-initialised) adapter parameters: keep the largest 10%."""
    ids = None if reset_params is None else {id(p) for p in reset_params}
    for group in optimizer.param_groups:
        for p in group["params"]:
            if ids is not None and id(p) not in ids:
                continue
            st = optimizer.state.get(p)
            if not st:
                continue
            for key in ("exp_avg", "exp_avg_sq"):
                if key in st:
                    t = st[key]
                    k = int(prune_ratio * t.numel())
                    if k > 0:
                        thresh = torch.kthvalue(t.abs().flatten().float(), k).values
                        st[key] = torch.where(t.abs() > thresh.to(t.dtype), t, torch.zeros_like(t))


def find_lora_modules(model) -> Dict[str, LoraLayer]:
    return {name: m for name, m in model.named_modules() if isinstance(m, LoraLayer)}


def lora_delta_weight(layer: LoraLayer, device=None) -> torch.Tensor:
    delta = None
    for ad in layer.active_adapters:
        a, b = layer.lora_A[ad].weight, layer.lora_B[ad].weight
        if device is not None:
            a, b = a.to(device), b.to(device)
        d = (b.float() @ a.float()) * layer.scaling[ad]
        delta = d if delta is None else delta + d
    return delta


@torch.no_grad()
def merge_and_reinit(model, reinit: bool = True):
    """Fold every adapter into its base layer (re-quantizing low-bit bases) and restart the adapters."""
    touched = []
    for name, layer in find_lora_modules(model).items():
        base = layer.base_layer
        delta = lora_delta_weight(layer)
        if isinstance(base, LowBitLinear):
            dev = base.weight.data.device
            w = dequant_weight(base.weight, torch.float32) + delta.to(dev)
            new = FP4Params(w.cpu(), requires_grad=False, quantized=False, qtype=base.qtype,
                            in_features=base.in_len).to("cpu")
            base.weight = new.to(dev) if dev.type != "cpu" else new
        else:
            base.weight.data.add_(delta.to(base.weight.dtype).to(base.weight.device))
        if reinit:
            for ad in layer.active_adapters:
                torch.nn.init.kaiming_uniform_(layer.lora_A[ad].weight, a=math.sqrt(5))
                torch.nn.init.zeros_(layer.lora_B[ad].weight)
                touched += [layer.lora_A[ad].weight, layer.lora_B[ad].weight]
    return touched


class ReLoRAScheduler(LRScheduler):
    """Wraps a scheduler: after every restart the rate climbs back from ``min_lr_scale`` over ``warmup_steps``."""

    def __init__(self, optimizer, inner_schedule: LRScheduler, relora_steps: int, warmup_steps: int,
                 min_lr_scale: float = 0.001) -> None:
        self.inner_schedule = inner_schedule
        self.relora_steps = relora_steps
        self.warmup_steps = warmup_steps
        self.min_lr_scale = min_lr_scale
        super().__init__(optimizer, inner_schedule.last_epoch)

    def get_lr(self):
        self.inner_schedule.last_epoch = self.last_epoch
        original = self.inner_schedule.get_lr()
        step = self.last_epoch
        if step < self.relora_steps:
            scale = 1.0
        else:
            cycle_t = min(1.0, (step % self.relora_steps) / max(1, self.warmup_steps))
            scale = cycle_t * (1 - self.min_lr_scale) + self.min_lr_scale
        return [lr * scale for lr in original]


class ReLoRACallback(TrainerCallback):
    def __init__(self, relora_steps=150, relora_cpu_offload=False, base_model="meta-llama/Llama-2-7b-hf",
                 resume_from_checkpoint=None):
        self.relora_steps = relora_steps
        self.cpu_offload = relora_cpu_offload
        self.last_full_model = base_model
        self.resume_from_checkpoint = resume_from_checkpoint
        self.num_merges = 0

    def on_step_begin(self, args, state, control, model=None, optimizer=None, **kwargs):
        if state.global_step > 0 and state.global_step % self.relora_steps == 0:
            touched = merge_and_reinit(model)
            if optimizer is not None:
                reset_optimizer(optimizer, touched)
            self.num_merges += 1
        return control

    def on_save(self, args, state, control, model=None, **kwargs):
        if is_main_process() and hasattr(model, "save_pretrained"):
            out = os.path.join(args.output_dir, f"checkpoint-{state.global_step}", "adapter")
            model.save_pretrained(out)
        return control

    def on_train_end(self, args, state, control, model=None, **kwargs):
        merge_and_reinit(model, reinit=False)
        return control


class ReLoRATrainer(Trainer):
    """HF ``Trainer`` with the ReLoRA restart schedule and merge callback installed."""

    def __init__(self, *args, base_model="meta-llama/Llama-2-7b-hf", relora_steps=150, relora_warmup_steps=10,
                 relora_cpu_offload=False, resume_from_checkpoint=None, **kwargs):
        self.lr_scheduler = None
        self.relora_steps = relora_steps
        self.relora_warmup_steps = relora_warmup_steps
        self.relora_cpu_offload = relora_cpu_offload
        callbacks = kwargs.get("callbacks", []) or []
        if self.relora_steps > 0:
            callbacks.append(ReLoRACallback(relora_steps, relora_cpu_offload, base_model, resume_from_checkpoint))
        kwargs["callbacks"] = callbacks
        super().__init__(*args, **kwargs)

    def create_scheduler(self, num_training_steps: int, optimizer: Optional[torch.optim.Optimizer] = None):
        optimizer = self.optimizer if optimizer is None else optimizer
        lr_scheduler = super().create_scheduler(num_training_steps, optimizer)
        if self.relora_steps:
            warmup = self.relora_warmup_steps if self.relora_warmup_steps else 10
            self.lr_scheduler = ReLoRAScheduler(optimizer, lr_scheduler, self.relora_steps, warmup)
        else:
            self.lr_scheduler = lr_scheduler
        return self.lr_scheduler
