"""A whole QLoRA / LoRA training step captured in one hipGraph.

A 7B QLoRA step at 4 x 256 tokens is about 3 500 kernel launches (seven adapted projections x 32 layers, forward and
backward, norms, rotary, attention, dropout, the optimizer): 80 ms of kernels that the host needs 95 ms to issue.  The
shapes of such a step never change, so forward + backward (+ the optimizer step when no gradient exchange sits between
them) are captured once and replayed: the host then issues ONE launch per step and the GPU runs the kernels back to
back.  The reference has no counterpart (its XPU runtime has no graphs; python/llm/example/GPU/LLM-Finetuning/QLoRA/
alpaca-qlora/alpaca_qlora_finetuning.py drives an eager ``transformers.Trainer``).

Usage::

    step = GraphedTrainStep(model, optimizer, example_ids)        # captures after 3 eager warm-up steps
    for ids in loader:                                            # every batch has example_ids' shape
        loss = step(ids)                                          # tensor, valid until the next call

Batches padded to one length pass ``labels`` (-100 on the padding) and ``attention_mask`` next to the ids; the example
batch given to the constructor decides which of them the captured step takes.

With a data-parallel group, gradient clipping or gradient accumulation the graph ends after backward; gradients are
averaged with the bucketed all-reduce of ``parallel/data_parallel.py`` and the optimizer runs eagerly (one fused launch
per bucket) after every ``grad_accum`` replays.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def capturable_adamw(params, **kw):
    """AdamW whose step holds no host-side state reads (step counts live on the device), as graph capture requires, in
    the one-launch multi-tensor form: the ``foreach`` form falls back to two launches per parameter for its
    bias-correction divisions when parameters are bf16 and step counts fp32 (896 launches, 4.7 ms of a Llama-2-7B QLoRA
    step with seven targets)."""
    params = list(params)
    if all(p.is_cuda and p.is_floating_point() for p in params):
        try:
            return torch.optim.AdamW(params, capturable=True, fused=True, **kw)
        except (RuntimeError, ValueError):
            pass
    return torch.optim.AdamW(params, capturable=True, foreach=True, **kw)


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_ids: torch.Tensor, *, example_labels: Optional[torch.Tensor] = None,
                 example_mask: Optional[torch.Tensor] = None, labels_fn: Optional[Callable] = None,
                 autocast_dtype=torch.bfloat16, group=None, warmup: int = 3, max_grad_norm: Optional[float] = None,
                 bucket_mb: int = 64, grad_accum: int = 1):
        if not example_ids.is_cuda:
            raise ValueError("graph capture needs the batch on the GPU")
        if any(getattr(m, "gradient_checkpointing", False) for m in model.modules()):
            # recomputation saves and restores the random-number state on the host, which a capture cannot contain
            raise ValueError("activation checkpointing cannot be captured: prepare the model with "
                             "use_gradient_checkpointing=False (a 7B QLoRA step at 4 x 256 tokens holds 3 GB of activations)")
        import torch.distributed as dist
        self.model, self.optimizer, self.group = model, optimizer, group
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.max_grad_norm, self.bucket_mb, self.grad_accum = max_grad_norm, bucket_mb, max(1, int(grad_accum))
        self.labels_fn = labels_fn or (lambda ids: ids)
        self.autocast_dtype = autocast_dtype
        self.params = [p for p in model.parameters() if p.requires_grad]
        # the optimizer step is inside the graph when nothing has to happen between backward and it
        self.opt_in_graph = not self.distributed and max_grad_norm is None and self.grad_accum == 1
        if self.opt_in_graph:
            for g in optimizer.param_groups:
                if "capturable" in g and not g["capturable"]:      # the Adam family reads its step count on the host
                    raise ValueError("the optimizer must be built with capturable=True (see capturable_adamw)")
        self.ids = example_ids.clone()
        self.labels = example_labels.clone() if example_labels is not None else None
        self.mask = example_mask.clone() if example_mask is not None else None
        self._micro = 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):           # allocator, workspace and autotune state settle before capture
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        if self.grad_accum > 1:
            # gradients exist (zero) before capture, so the captured backward ADDS to them; they are zeroed in place
            # after each optimizer step
            for p in self.params:
                p.grad = torch.zeros_like(p)
        else:
            optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._forward_backward()
            if self.opt_in_graph:
                optimizer.step()
        # gradients allocated during capture live in the graph's pool: replays overwrite them in place, so they are
        # never set to None again
        if self.grad_accum > 1:
            for p in self.params:             # the capture itself ran no kernels; start from zero
                p.grad.zero_()

    def _forward_backward(self):
        with torch.autocast("cuda", dtype=self.autocast_dtype):
            kw = {} if self.mask is None else {"attention_mask": self.mask}
            out = self.model(input_ids=self.ids, labels=self.labels if self.labels is not None else self.labels_fn(self.ids),
                             **kw)
        loss = out.loss / self.grad_accum if self.grad_accum > 1 else out.loss
        loss.backward()
        return out.loss.detach()

    def _finish_eager(self):
        if self.distributed:
            from ..parallel.data_parallel import sync_gradients
            sync_gradients(self.model, self.group, self.bucket_mb)
        if self.max_grad_norm:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
        self.optimizer.step()

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self._forward_backward()
        self._finish_eager()
        return loss

    def __call__(self, ids: torch.Tensor, labels: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One micro-batch; returns its loss (a tensor that the next call overwrites).  ``stepped`` tells whether this
        call ended with an optimizer step."""
        if ids.shape != self.ids.shape:
            raise ValueError(f"captured for batches of shape {tuple(self.ids.shape)}, got {tuple(ids.shape)}")
        if (labels is None) != (self.labels is None) or (attention_mask is None) != (self.mask is None):
            raise ValueError("pass labels / attention_mask exactly when the example batch had them")
        self.ids.copy_(ids, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        if attention_mask is not None:
            self.mask.copy_(attention_mask, non_blocking=True)
        self.graph.replay()
        self._micro += 1
        self.stepped = self.opt_in_graph or self._micro % self.grad_accum == 0
        if not self.opt_in_graph and self.stepped:
            self._finish_eager()
            if self.grad_accum > 1:
                for p in self.params:
                    p.grad.zero_()
        return self.loss
