"""Direct Preference Optimization on a low-bit base model with LoRA adapters.

The reference's DPO example (python/llm/example/GPU/LLM-Finetuning/DPO/dpo_finetuning.py) hands an nf4 model with
LoRA adapters to ``trl.DPOTrainer`` with ``ref_model=None``: the reference policy is the same network with the adapters
switched off.  This module is that recipe without trl: one forward of chosen + rejected through the policy, one
no-grad forward with the adapters disabled, and the sigmoid / hinge / IPO preference losses.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


def sequence_logps(logits: torch.Tensor, labels: torch.Tensor, average: bool = False) -> torch.Tensor:
    """Sum (or mean) over the response tokens of log p(label).  labels: [B, S] with -100 on prompt / padding."""
    logits, labels = logits[:, :-1].float(), labels[:, 1:]
    mask = labels != -100
    tok = torch.gather(logits.log_softmax(-1), 2, labels.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    s = (tok * mask).sum(-1)
    return s / mask.sum(-1).clamp(min=1) if average else s


def dpo_loss(policy_chosen, policy_rejected, ref_chosen, ref_rejected, beta: float = 0.1, loss_type: str = "sigmoid",
             label_smoothing: float = 0.0):
    """-> (loss per pair, chosen reward, rejected reward); rewards are beta * (log pi - log pi_ref), detached."""
    logits = (policy_chosen - policy_rejected) - (ref_chosen - ref_rejected)
    if loss_type == "sigmoid":
        loss = -F.logsigmoid(beta * logits) * (1 - label_smoothing) - F.logsigmoid(-beta * logits) * label_smoothing
    elif loss_type == "hinge":
        loss = torch.relu(1 - beta * logits)
    elif loss_type == "ipo":
        loss = (logits - 1 / (2 * beta)) ** 2
    else:
        raise ValueError(f"unknown loss_type {loss_type}")
    return loss, beta * (policy_chosen - ref_chosen).detach(), beta * (policy_rejected - ref_rejected).detach()


def build_pair_batch(tokenizer, rows: Sequence[Dict[str, str]], max_length: int = 512, max_prompt_length: int = 256):
    """rows of {prompt, chosen, rejected} -> padded tensors: chosen and rejected stacked along the batch, labels -100
    on the prompt tokens and the padding."""
    seqs, labels = [], []
    for key in ("chosen", "rejected"):
        for r in rows:
            p = tokenizer(r["prompt"], add_special_tokens=True)["input_ids"][-max_prompt_length:]
            a = tokenizer(r[key], add_special_tokens=False)["input_ids"]
            if tokenizer.eos_token_id is not None:
                a = a + [tokenizer.eos_token_id]
            ids = (p + a)[:max_length]
            seqs.append(ids)
            labels.append(([-100] * len(p) + a)[:max_length])
    width = max(len(s) for s in seqs)
    pad = tokenizer.pad_token_id if tokenizer.pad_token_id is not None else 0
    ids = torch.tensor([s + [pad] * (width - len(s)) for s in seqs])
    lab = torch.tensor([l + [-100] * (width - len(l)) for l in labels])
    att = torch.tensor([[1] * len(s) + [0] * (width - len(s)) for s in seqs])
    return {"input_ids": ids, "labels": lab, "attention_mask": att}


@dataclass
class DPOConfig:
    beta: float = 0.1
    loss_type: str = "sigmoid"
    label_smoothing: float = 0.0
    learning_rate: float = 5e-5
    max_grad_norm: float = 0.3
    gradient_accumulation_steps: int = 1
    bf16: bool = True          # on the GPU the forwards run under bf16 autocast (the reference's TrainingArguments force bf16)


class DPOTrainer:
    """``DPOTrainer(peft_model, DPOConfig(...), ref_model=None)``; ``step(batch)`` takes the output of
    ``build_pair_batch`` (chosen rows first, rejected rows second)."""

    def __init__(self, model, config: Optional[DPOConfig] = None, ref_model=None, optimizer=None):
        self.model, self.cfg, self.ref_model = model, config or DPOConfig(), ref_model
        if ref_model is None and not hasattr(model, "disable_adapter"):
            raise ValueError("without ref_model the policy must be an adapter model (disable_adapter())")
        params = [p for p in model.parameters() if p.requires_grad]
        self.optimizer = optimizer or torch.optim.AdamW(params, lr=self.cfg.learning_rate)
        self._micro = 0
        self.history: List[Dict[str, float]] = []

    def _logps(self, model, batch):
        ids = batch["input_ids"]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.cfg.bf16 and ids.device.type == "cuda"):
            out = model(input_ids=ids, attention_mask=batch.get("attention_mask"))
        return sequence_logps(out.logits, batch["labels"], average=self.cfg.loss_type == "ipo")

    def reference_logps(self, batch):
        with torch.no_grad():
            if self.ref_model is not None:
                return self._logps(self.ref_model, batch)
            with self.model.disable_adapter():
                return self._logps(self.model, batch)

    def step(self, batch) -> Dict[str, float]:
        n = batch["input_ids"].shape[0] // 2
        ref = self.reference_logps(batch)
        pol = self._logps(self.model, batch)
        loss, rc, rr = dpo_loss(pol[:n], pol[n:], ref[:n], ref[n:], self.cfg.beta, self.cfg.loss_type,
                                self.cfg.label_smoothing)
        (loss.mean() / self.cfg.gradient_accumulation_steps).backward()
        self._micro += 1
        if self._micro % self.cfg.gradient_accumulation_steps == 0:
            if self.cfg.max_grad_norm:
                torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad],
                                               self.cfg.max_grad_norm)
            self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)
        m = {"loss": loss.mean().item(), "rewards/chosen": rc.mean().item(), "rewards/rejected": rr.mean().item(),
             "rewards/accuracies": (rc > rr).float().mean().item(), "rewards/margins": (rc - rr).mean().item()}
        self.history.append(m)
        return m
