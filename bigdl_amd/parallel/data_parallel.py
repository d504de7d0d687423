"""Data parallelism for adapter fine-tuning: every rank holds the frozen low-bit base model and a replica of the
(small) trainable parameters; gradients are summed in a few flat buckets after backward.

The reference runs QLoRA data-parallel through ``accelerate`` / DeepSpeed ZeRO-2 over oneCCL
(python/llm/example/GPU/LLM-Finetuning/QLoRA/alpaca-qlora/deepspeed_zero2.json and
``training_patch.py``).  LoRA gradients are a few tens of MB, so a ZeRO-style partition buys nothing on 288 GB
GPUs: a plain bucketed all-reduce (one or two RCCL calls per step) is the cheapest correct scheme.  HF ``Trainer``
users get the same effect from torch DDP, which works unchanged on RCCL.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .comm import all_reduce_coalesced_, world_size


def broadcast_trainable(model, src: int = 0, group=None):
    """Make every replica start from rank ``src``'s adapter weights."""
    if world_size() <= 1:
        return
    for p in model.parameters():
        if p.requires_grad:
            dist.broadcast(p.data, src=src, group=group)


def sync_gradients(model, group=None, bucket_mb: int = 64):
    """Average the gradients of the trainable parameters across the data-parallel group."""
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    all_reduce_coalesced_(grads, group=group, bucket_mb=bucket_mb, average=True)
    return grads


class DataParallelAdapter:
    """Minimal wrapper: ``step(loss)`` = backward + gradient all-reduce + optimizer step."""

    def __init__(self, model, optimizer, group=None, bucket_mb: int = 64, grad_accum: int = 1, max_grad_norm=None):
        self.model, self.optimizer, self.group = model, optimizer, group
        self.bucket_mb, self.grad_accum, self.max_grad_norm = bucket_mb, grad_accum, max_grad_norm
        self._micro = 0
        broadcast_trainable(model, 0, group)

    def step(self, loss: torch.Tensor):
        (loss / self.grad_accum).backward()
        self._micro += 1
        if self._micro % self.grad_accum:
            return False
        sync_gradients(self.model, self.group, self.bucket_mb)
        if self.max_grad_norm:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], self.max_grad_norm)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        return True
