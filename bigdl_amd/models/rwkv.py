"""RWKV-4 and RWKV-5 forwards on the recurrence kernels (reference: python/llm/src/ipex_llm/transformers/models/rwkv4.py
``rwkv_attention_forward`` :104-181 with ``rwkv_time_shift`` + ``rwkv_linear_attention_v4``; rwkv5.py ``extract_key_value``
:41-68, ``rwkv_linear_attention_xpu`` :121-164, ``rwkv_attention_forward`` :167-216, ``rwkv_ffn_forward`` :219-254).

RWKV-5 ("world") models ship as remote code: the classes are recognised by name (``Rwkv5SelfAttention`` /
``Rwkv5FeedForward``) and by their parameters (``time_faaaa``, ``ln_x``, four ``time_mix_*`` vectors).  The recurrent
state keeps the layout of the model's own code -- per-layer slices along the LAST dimension:
state[0] [B, C, L] attention shift, state[1] [B, H, S, S, L] wkv matrices, state[2] [B, C, L] feed-forward shift --
so that ``generate`` and the model forward need no wrapper; a layer's [B, H, S, S] matrix is copied into a contiguous
fp32 scratch for the kernel and written back (64 KB per head at S = 64).
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


def rwkv5_linear_attention(B, H, S, T, hidden, time_decay, time_first, receptance, key, value, gate, lxw, lxb, ow, state,
                           eps=1e-5):
    """RWKV-5 time mixing tail (reference rwkv5.py:70-118): wkv with a per-head [S, S] state, group norm, gate,
    output projection.  ``state`` [B, H, S, S] fp32 is updated in place."""
    from ..ops import hip
    key = key.float().view(B, T, H, S).transpose(1, 2)
    value = value.float().view(B, T, H, S).transpose(1, 2)
    receptance = receptance.float().view(B, T, H, S).transpose(1, 2)
    w = torch.exp(-torch.exp(time_decay.float()))
    out = hip.rwkv_linear_attention_v5(w, time_first.float(), receptance, key, value, state)
    out = out.reshape(B * T, H * S)
    out = F.group_norm(out, num_groups=H, weight=lxw.float(), bias=lxb.float(), eps=eps).reshape(B, T, H * S)
    return ow(out.to(dtype=hidden.dtype) * gate)


def _shifted(self, hidden, state, slot):
    if hidden.size(1) == 1 and state is not None:
        return state[slot][:, :, self.layer_id].unsqueeze(1)
    shifted = self.time_shift(hidden)
    if state is not None:
        shifted[:, 0] = state[slot][:, :, self.layer_id]
    return shifted


def _mix(self, names, hidden):
    mix = getattr(self, "_mixed_mix", None)
    if mix is None or mix.dtype != hidden.dtype or mix.device != hidden.device:
        mix = torch.cat([getattr(self, n).data.reshape(-1) for n in names]).reshape(len(names), -1).to(hidden.dtype)
        object.__setattr__(self, "_mixed_mix", mix)
    return mix


def rwkv5_attention_forward(self, hidden, state=None, use_cache=False, seq_mode=True):
    """Drop-in for the remote code's ``Rwkv5SelfAttention.forward``: returns (output, state)."""
    if hidden.device.type != "cuda" or torch.is_grad_enabled():
        return self._hf_forward(hidden, state=state, use_cache=use_cache, seq_mode=seq_mode)
    from ..ops import hip
    B, T, C = hidden.shape
    H = self.time_decay.shape[0]
    S = C // H
    shifted = _shifted(self, hidden, state, 0)
    mix = _mix(self, ("time_mix_key", "time_mix_value", "time_mix_receptance", "time_mix_gate"), hidden)
    k_in, v_in, r_in, g_in = hip.rwkv_time_shift(hidden, shifted.to(hidden.dtype), mix)
    key, value, receptance = self.key(k_in), self.value(v_in), self.receptance(r_in)
    gate = F.silu(self.gate(g_in))
    if state is not None:
        state[0][:, :, self.layer_id] = hidden[:, -1]
        layer_state = state[1][:, :, :, :, self.layer_id].float().contiguous()
    else:
        layer_state = torch.zeros(B, H, S, S, dtype=torch.float32, device=hidden.device)
    out = rwkv5_linear_attention(B, H, S, T, hidden, self.time_decay, self.time_faaaa, receptance, key, value, gate,
                                 self.ln_x.weight, self.ln_x.bias, self.output, layer_state)
    if state is not None:
        state[1][:, :, :, :, self.layer_id] = layer_state.to(state[1].dtype)
    return out, state


def rwkv5_ffn_forward(self, hidden, state=None):
    """Drop-in for ``Rwkv5FeedForward.forward`` (channel mixing): returns (output, state)."""
    if hidden.device.type != "cuda" or torch.is_grad_enabled():
        return self._hf_forward(hidden, state=state)
    from ..ops import hip
    shifted = _shifted(self, hidden, state, 2)
    mix = _mix(self, ("time_mix_key", "time_mix_receptance"), hidden)
    k_in, r_in = hip.rwkv_time_shift(hidden, shifted.to(hidden.dtype), mix)
    key = torch.square(torch.relu(self.key(k_in)))
    value = self.value(key)
    receptance = torch.sigmoid(self.receptance(r_in))
    if state is not None:
        state[2][:, :, self.layer_id] = hidden[:, -1]
    return receptance * value, state


def bind_rwkv(model):
    """Rebind the time-mixing (and, for RWKV-5, channel-mixing) modules of a HuggingFace RWKV-4 / remote-code RWKV-5
    model."""
    import types
    for m in model.modules():
        name = m.__class__.__name__
        if hasattr(m, "_hf_forward"):
            continue
        if name == "RwkvSelfAttention":
            fwd = rwkv_attention_forward
        elif name == "Rwkv5SelfAttention" and hasattr(m, "time_faaaa") and hasattr(m, "ln_x"):
            fwd = rwkv5_attention_forward
        elif name == "Rwkv5FeedForward":
            fwd = rwkv5_ffn_forward
        else:
            continue
        object.__setattr__(m, "_hf_forward", m.forward)
        m.forward = types.MethodType(fwd, m)
    return model
