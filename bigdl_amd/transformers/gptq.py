"""Containers for GPTQ-packed linears so that an AutoGPTQ / optimum checkpoint loads without those packages.

The reference relies on ``auto_gptq`` / ``optimum`` being installed to build ``QuantLinear`` modules before it
re-blocks them (python/llm/src/ipex_llm/transformers/model.py:251-295, convert.py:379-453).  Here a plain module with
the same buffers (``qweight`` [in/8, out] int32, ``qzeros`` [groups, out/8] int32, ``scales`` [groups, out] fp16,
``g_idx`` [in] int32, optional ``bias``) receives the checkpoint; ``convert_gptq`` (convert.py) then turns it into the
asym_int4 block layout.  The container has no forward of its own: it only exists between loading and conversion.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from ..utils.common import invalidInputError


class GPTQQuantLinear(nn.Module):
    def __init__(self, bits, group_size, in_features, out_features, bias, dev="cpu"):
        super().__init__()
        invalidInputError(bits in (2, 3, 4, 8), "GPTQ bits must be 2, 3, 4 or 8")
        self.bits = bits
        self.group_size = group_size if group_size != -1 else in_features
        self.infeatures = self.in_features = in_features
        self.outfeatures = self.out_features = out_features
        groups = math.ceil(in_features / self.group_size)
        self.register_buffer("qweight", torch.zeros((in_features // 32 * bits, out_features), dtype=torch.int32, device=dev))
        self.register_buffer("qzeros", torch.zeros((groups, out_features // 32 * bits), dtype=torch.int32, device=dev))
        self.register_buffer("scales", torch.zeros((groups, out_features), dtype=torch.float16, device=dev))
        self.register_buffer("g_idx", torch.tensor([i // self.group_size for i in range(in_features)], dtype=torch.int32,
                                                   device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float16, device=dev))
        else:
            self.bias = None

    def forward(self, x):
        invalidInputError(False, "GPTQQuantLinear only carries checkpoint tensors; convert the model with "
                                 "load_in_low_bit='asym_int4' before running it")


def replace_with_gptq_layers(model, q_config, modules_to_not_convert=None, prefix=""):
    bits = q_config.get("bits", 4)
    group_size = q_config.get("group_size", 128)
    skip = list(modules_to_not_convert or q_config.get("modules_to_not_convert") or []) + ["lm_head"]
    for name, module in model.named_children():
        full = f"{prefix}.{name}" if prefix else name
        if isinstance(module, nn.Linear) and not any(s in full for s in skip):
            model._modules[name] = GPTQQuantLinear(bits, group_size, module.in_features, module.out_features,
                                                   module.bias is not None, dev=module.weight.device)
        elif len(list(module.children())) > 0:
            replace_with_gptq_layers(module, q_config, modules_to_not_convert, full)
    return model


def pack_gptq(weight: torch.Tensor, group_size: int = 128, bits: int = 4, sym: bool = False):
    """Round-to-nearest quantize a dense [out, in] weight into GPTQ tensors (qweight, qzeros, scales, g_idx).
    Used to export to the GPTQ layout and to build test fixtures."""
    out_f, in_f = weight.shape
    groups = in_f // group_size
    w = weight.float().view(out_f, groups, group_size)
    maxq = 2 ** bits - 1
    wmax, wmin = w.amax(-1), w.amin(-1)
    if sym:
        m = torch.maximum(wmax.abs(), wmin.abs())
        wmax, wmin = m, -m
    scale = ((wmax - wmin) / maxq).clamp_min(1e-8)
    zero = torch.round(-wmin / scale).clamp(0, maxq)
    q = torch.clamp(torch.round(w / scale[..., None]) + zero[..., None], 0, maxq).to(torch.int32).view(out_f, in_f)
    per = 32 // bits
    qt = q.t().contiguous().view(in_f // per, per, out_f)                     # fields along the input dim
    shifts = (torch.arange(per, dtype=torch.int64) * bits).view(1, per, 1)
    qweight = (qt.to(torch.int64) << shifts).sum(1)
    z = (zero.t().contiguous().to(torch.int64) - 1) & maxq                    # stored as zero - 1
    zt = z.view(groups, out_f // per, per)
    qzeros = (zt << (torch.arange(per, dtype=torch.int64) * bits).view(1, 1, per)).sum(-1)
    to_i32 = lambda t: torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)  # noqa: E731
    g_idx = torch.tensor([i // group_size for i in range(in_f)], dtype=torch.int32)
    return to_i32(qweight), to_i32(qzeros), scale.t().contiguous().to(torch.float16), g_idx
