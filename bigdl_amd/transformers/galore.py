"""GaLore: memory-efficient full-parameter training by projecting each 2-D gradient onto a rank-r subspace, running
Adam there, and projecting the update back (Zhao et al. 2024).

The reference's example (python/llm/example/GPU/LLM-Finetuning/GaLore/galore_finetuning.py) selects the attention /
MLP matrices with ``optim_target_modules=["attn", "mlp"]`` and uses ``galore_adamw``; this is the optimizer itself:
``GaLoreAdamW(galore_param_groups(model, ["attn", "mlp"], rank=128, ...), lr=...)``.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence

import torch


def galore_param_groups(model, target_modules: Sequence[str], rank: int = 128, update_proj_gap: int = 200,
                        scale: float = 0.25, proj_type: str = "std"):
    """Two parameter groups: the 2-D weights of modules whose name contains one of ``target_modules`` (projected),
    and everything else trainable (plain AdamW)."""
    low, rest = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.dim() == 2 and any(t in name for t in target_modules):
            low.append(p)
        else:
            rest.append(p)
    return [{"params": rest},
            {"params": low, "rank": rank, "update_proj_gap": update_proj_gap, "scale": scale, "proj_type": proj_type}]


class _Projector:
    def __init__(self, rank, update_proj_gap, scale, proj_type):
        self.rank, self.gap, self.scale, self.proj_type = rank, update_proj_gap, scale, proj_type
        self.ortho = None
        self.left = True

    def _refresh(self, g):
        m, n = g.shape
        self.left = m <= n if self.proj_type == "std" else self.proj_type == "left"
        u, _, vh = torch.linalg.svd(g.float(), full_matrices=False)
        r = min(self.rank, u.shape[1])
        self.ortho = (u[:, :r] if self.left else vh[:r]).to(g.dtype).contiguous()

    def project(self, g, step):
        if self.ortho is None or step % self.gap == 0:
            self._refresh(g)
        return self.ortho.t() @ g if self.left else g @ self.ortho.t()

    def project_back(self, u):
        return (self.ortho @ u if self.left else u @ self.ortho) * self.scale


class GaLoreAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if "step" not in st:
                    st["step"] = 0
                if "rank" in group and g.dim() == 2:
                    if "projector" not in st:
                        st["projector"] = _Projector(group["rank"], group["update_proj_gap"], group["scale"],
                                                     group["proj_type"])
                    g = st["projector"].project(g, st["step"])
                if "exp_avg" not in st or st["exp_avg"].shape != g.shape:
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(g), torch.zeros_like(g)
                st["step"] += 1
                st["exp_avg"].mul_(b1).add_(g, alpha=1 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = st["exp_avg_sq"].sqrt().add_(group["eps"])
                step_size = group["lr"] * math.sqrt(1 - b2 ** st["step"]) / (1 - b1 ** st["step"])
                upd = st["exp_avg"] / denom
                if "projector" in st:
                    upd = st["projector"].project_back(upd)
                if group["weight_decay"]:
                    p.mul_(1 - group["lr"] * group["weight_decay"])
                p.add_(upd.to(p.dtype), alpha=-step_size)
        return loss

    def state_numel(self) -> int:
        """Elements of optimizer state (moments + projection bases): what GaLore saves against AdamW's 2 * numel."""
        n = 0
        for st in self.state.values():
            n += st["exp_avg"].numel() + st["exp_avg_sq"].numel()
            if "projector" in st and st["projector"].ortho is not None:
                n += st["projector"].ortho.numel()
        return n
