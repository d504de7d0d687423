"""Autograd nodes for the element-wise parts of a decoder layer on the GPU training path (QLoRA / LoRA fine-tuning
through the HuggingFace modules).

In training the bound forwards used to fall back to the models' own PyTorch expressions: an RMSNorm is seven small
kernels forward (cast, square, mean, rsqrt, two multiplies, cast back) and about twice that backward, ``silu(g) * u`` is
two forward and four backward, rotary embedding a chain of slices, concatenations and multiply-adds per tensor.  On a
Llama-2-7B QLoRA step that is a fifth of the kernel time and most of the launches.  Each node here is one kernel
forward and one backward (``csrc/hip/misc_ops.hip``: ``rms_norm_kernel`` / ``rms_norm_bwd_kernel``, ``act_mul_kernel`` /
``silu_mul_bwd_kernel``, ``rope_kernel`` both ways -- the transpose of a rotation is the rotation by the opposite
angle).  The reference gets the same effect from its fused XPU ops in the training forwards
(python/llm/src/ipex_llm/transformers/models/llama.py, ``rms_norm_forward`` / ``mlp_silu_forward`` / fused rope).
"""
from __future__ import annotations

import torch


class RMSNormFn(torch.autograd.Function):
    """y = x * rsqrt(mean(x^2) + eps) * (weight + offset)"""

    @staticmethod
    def forward(ctx, x, weight, eps, offset):
        from ..ops import hip
        ctx.save_for_backward(x, weight)
        ctx.eps, ctx.offset = eps, offset
        return hip.rms_norm(x, weight, eps, offset=offset)

    @staticmethod
    def backward(ctx, gy):
        from ..ops import hip
        x, weight = ctx.saved_tensors
        dx = hip.rms_norm_bwd(x, weight, gy, ctx.eps, ctx.offset) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:                   # (full fine-tuning: norms are frozen under LoRA)
            xf = x.float()
            rs = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + ctx.eps)
            dw = (gy.float() * xf * rs).reshape(-1, x.shape[-1]).sum(0).to(weight.dtype)
        return dx, dw, None, None


class SiluMulFn(torch.autograd.Function):
    """out = silu(gate) * up"""

    @staticmethod
    def forward(ctx, gate, up):
        from ..ops import hip
        ctx.save_for_backward(gate, up)
        out = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
        return hip.act_mul(gate, up, "silu", out=out)

    @staticmethod
    def backward(ctx, g):
        from ..ops import hip
        gate, up = ctx.saved_tensors
        return hip.silu_mul_bwd(gate, up, g)


class RopeFn(torch.autograd.Function):
    """Rotary embedding of a [B, H, S, D] tensor with HuggingFace cos / sin tables [B, S, rot] (half-split pairs)."""

    @staticmethod
    def forward(ctx, x, cos, sin):
        from ..ops import hip
        ctx.save_for_backward(cos, sin)
        y = x.clone(memory_format=torch.contiguous_format)
        hip.rope_inplace(y, cos=cos, sin=sin, rot_dim=cos.shape[-1])
        return y

    @staticmethod
    def backward(ctx, g):
        from ..ops import hip
        cos, sin = ctx.saved_tensors
        gx = g.clone(memory_format=torch.contiguous_format)
        hip.rope_inplace(gx, cos=cos, sin=-sin, rot_dim=cos.shape[-1])
        return gx, None, None


def usable(*tensors) -> bool:
    """GPU tensors of a dtype the kernels take, in a pass that records gradients"""
    return torch.is_grad_enabled() and all(t.is_cuda and t.dtype in (torch.float16, torch.bfloat16, torch.float32) for t in tensors)


def bind_fused_training_ops(model) -> int:
    """Point the element-wise pieces of ``model``'s decoder layers at the nodes above, whatever ``optimize_model`` did
    at load time (fine-tuning recipes load with ``optimize_model=False``): RMSNorm classes whose forward is the plain
    formula, gated MLPs that are ``down(act(gate(x)) * up(x))`` and nothing else, and the modelling file's
    ``apply_rotary_pos_emb``.  Each bound forward still falls back to the module's PyTorch expression when the node does
    not apply (CPU, no gradient, other activation).  Returns the number of classes / functions rebound."""
    import sys
    from ..models import hf_patch as P
    from .convert import convert_forward
    n, seen = 0, set()
    for m in model.modules():
        cls = type(m)
        if cls in seen:
            continue
        if cls.__name__.endswith("RMSNorm") and getattr(m, "weight", None) is not None and m.weight.dim() == 1:
            seen.add(cls)
            if getattr(m.forward, "__func__", None) is not P.rms_norm_forward and P._rms_forward_matches(m):
                convert_forward(model, cls, P.rms_norm_forward)
                n += 1
        elif all(hasattr(m, a) for a in ("gate_proj", "up_proj", "down_proj", "act_fn")) and not hasattr(m, "num_experts"):
            seen.add(cls)
            if getattr(m.forward, "__func__", None) is not P.mlp_forward and P._plain_gated_mlp(m):
                convert_forward(model, cls, P.mlp_forward)
                n += 1
        elif cls.__name__.endswith("Attention") and cls.__module__ not in seen:
            seen.add(cls.__module__)
            mod = sys.modules.get(cls.__module__)
            own = getattr(mod, "apply_rotary_pos_emb", None)
            if own is not None and not getattr(own, "_bigdl_fused", False) and _is_half_split_rope(own):
                def fused(q, k, cos, sin, position_ids=None, unsqueeze_dim=1, _own=own):
                    if unsqueeze_dim == 1 and cos.dim() == 3 and q.dim() == 4 and cos.shape[-1] == q.shape[-1]:
                        return P.apply_rotary(q, k, cos, sin)
                    return _own(q, k, cos, sin, position_ids, unsqueeze_dim) if position_ids is not None else _own(q, k, cos, sin, unsqueeze_dim=unsqueeze_dim)
                fused._bigdl_fused = True
                mod.apply_rotary_pos_emb = fused
                n += 1
    return n


def _is_half_split_rope(fn) -> bool:
    """Does the modelling file's rotary function compute q cos + rotate_half(q) sin on [B, H, S, D] with [B, S, D] tables?"""
    from ..models.hf_patch import apply_rotary
    try:
        q, k = torch.randn(1, 2, 3, 8), torch.randn(1, 1, 3, 8)
        ang = torch.randn(1, 3, 4)
        cos, sin = torch.cat([ang.cos()] * 2, -1), torch.cat([ang.sin()] * 2, -1)
        with torch.no_grad():
            a = fn(q, k, cos, sin)
            b = apply_rotary(q, k, cos, sin)
        return torch.allclose(a[0], b[0], atol=1e-5) and torch.allclose(a[1], b[1], atol=1e-5)
    except Exception:
        return False
