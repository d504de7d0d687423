"""RWKV-4 attention forward on the recurrence kernels (reference: python/llm/src/ipex_llm/transformers/models/rwkv4.py:
``rwkv_attention_forward`` :104-181 with ``rwkv_time_shift`` + ``rwkv_linear_attention_v4``; rwkv5.py has the same
structure with the matrix-state kernel, exposed here as ``rwkv5_linear_attention`` for models shipped with remote code).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def rwkv_attention_forward(self, hidden, state=None, use_cache=False):
    """Drop-in for HuggingFace ``RwkvSelfAttention.forward``."""
    if hidden.device.type != "cuda" or torch.is_grad_enabled():
        return self._hf_forward(hidden, state=state, use_cache=use_cache)
    from ..ops import hip
    if hidden.size(1) == 1 and state is not None:
        shifted = state[1][:, :, self.layer_id].unsqueeze(1)
    else:
        shifted = self.time_shift(hidden)
        if state is not None:
            shifted[:, 0] = state[1][:, :, self.layer_id]
    mix = getattr(self, "_mixed_mix", None)
    if mix is None or mix.dtype != hidden.dtype or mix.device != hidden.device:
        mix = torch.cat([self.time_mix_key.data, self.time_mix_value.data, self.time_mix_receptance.data]).reshape(
            3, -1).to(hidden.dtype)
        object.__setattr__(self, "_mixed_mix", mix)
    k_in, v_in, r_in = hip.rwkv_time_shift(hidden, shifted.to(hidden.dtype), mix)
    key, value = self.key(k_in), self.value(v_in)
    receptance = torch.sigmoid(self.receptance(r_in))
    if state is not None:
        state[1][:, :, self.layer_id] = hidden[:, -1]
    if state is not None:
        num, den, mx = (state[i][:, :, self.layer_id].float().contiguous() for i in (2, 3, 4))
    elif use_cache:
        b, _, c = key.shape
        num = torch.zeros(b, c, dtype=torch.float32, device=key.device)
        den = torch.zeros_like(num)
        mx = torch.full_like(num, -1e38)
    else:
        num = den = mx = None
    rwkv = hip.rwkv_linear_attention_v4(self.time_decay, self.time_first, key, value, num, den, mx)
    if state is not None:
        state[2][:, :, self.layer_id] = num
        state[3][:, :, self.layer_id] = den
        state[4][:, :, self.layer_id] = mx
    return self.output(receptance * rwkv), state


def rwkv5_linear_attention(B, H, S, T, hidden, time_decay, time_first, receptance, key, value, gate, lxw, lxb, ow, state):
    """RWKV-5 time mixing tail (reference rwkv5.py:70-118): wkv with a per-head [S, S] state, group norm, gate,
    output projection.  ``state`` [B, H, S, S] fp32 is updated in place."""
    from ..ops import hip
    key = key.float().view(B, T, H, S).transpose(1, 2)
    value = value.float().view(B, T, H, S).transpose(1, 2)
    receptance = receptance.float().view(B, T, H, S).transpose(1, 2)
    w = torch.exp(-torch.exp(time_decay.float()))
    out = hip.rwkv_linear_attention_v5(w, time_first.float(), receptance, key, value, state)
    out = out.reshape(B * T, H * S)
    out = F.group_norm(out, num_groups=H, weight=lxw.float(), bias=lxb.float()).reshape(B, T, H * S)
    return ow(out.to(dtype=hidden.dtype) * gate)


def bind_rwkv(model):
    """Rebind every ``RwkvSelfAttention`` of a HuggingFace RWKV-4 model."""
    import types
    for m in model.modules():
        if m.__class__.__name__ == "RwkvSelfAttention" and not hasattr(m, "_hf_forward"):
            object.__setattr__(m, "_hf_forward", m.forward)
            m.forward = types.MethodType(rwkv_attention_forward, m)
    return model
