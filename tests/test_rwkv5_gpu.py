"""RWKV-5 binding (bigdl_amd/models/rwkv.py) against a plain-torch module that follows the published RWKV-5 time-mixing
/ channel-mixing recurrences (the same formulas as the reference's CPU path, models/rwkv5.py:121-160 and :219-254).  The
real model classes ship as remote code, which is not available offline: the stand-ins below carry the same class names,
parameter names and state layout ([..., layer] slices), which is what ``bind_rwkv`` keys on."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class Rwkv5SelfAttention(nn.Module):
    def __init__(self, hidden, heads, layer_id):
        super().__init__()
        self.layer_id, self.num_attention_heads = layer_id, heads
        s = hidden // heads
        self.time_decay = nn.Parameter(torch.randn(heads, s) * 0.5 - 1.0)
        self.time_faaaa = nn.Parameter(torch.randn(heads, s) * 0.3)
        for n in ("key", "value", "receptance", "gate"):
            setattr(self, f"time_mix_{n}", nn.Parameter(torch.rand(1, 1, hidden)))
            setattr(self, n, nn.Linear(hidden, hidden, bias=False))
        self.output = nn.Linear(hidden, hidden, bias=False)
        self.ln_x = nn.GroupNorm(heads, hidden)
        self.time_shift = nn.ZeroPad2d((0, 0, 1, -1))

    def forward(self, hidden, state=None, use_cache=False, seq_mode=True):
        B, T, C = hidden.shape
        H = self.time_decay.shape[0]
        S = C // H
        if T == 1 and state is not None:
            shifted = state[0][:, :, self.layer_id].unsqueeze(1)
        else:
            shifted = self.time_shift(hidden)
            if state is not None:
                shifted[:, 0] = state[0][:, :, self.layer_id]
        mixed = {n: hidden * getattr(self, f"time_mix_{n}") + shifted * (1 - getattr(self, f"time_mix_{n}"))
                 for n in ("key", "value", "receptance", "gate")}
        k = self.key(mixed["key"]).float().view(B, T, H, S).transpose(1, 2).transpose(-2, -1)
        v = self.value(mixed["value"]).float().view(B, T, H, S).transpose(1, 2)
        r = self.receptance(mixed["receptance"]).float().view(B, T, H, S).transpose(1, 2)
        g = F.silu(self.gate(mixed["gate"]))
        if state is not None:
            state[0][:, :, self.layer_id] = hidden[:, -1]
        st = state[1][:, :, :, :, self.layer_id].float() if state is not None else torch.zeros(B, H, S, S, device=hidden.device)
        w = torch.exp(-torch.exp(self.time_decay.float())).reshape(H, S, 1)
        u = self.time_faaaa.float().reshape(H, S, 1)
        out = torch.zeros(B, T, H, S, device=hidden.device)
        for t in range(T):
            at = k[:, :, :, t:t + 1] @ v[:, :, t:t + 1, :]
            out[:, t] = (r[:, :, t:t + 1, :] @ (u * at + st)).squeeze(2)
            st = at + w * st
        if state is not None:
            state[1][:, :, :, :, self.layer_id] = st
        out = F.group_norm(out.reshape(B * T, C), H, self.ln_x.weight.float(), self.ln_x.bias.float()).reshape(B, T, C)
        return self.output(out.to(hidden.dtype) * g), state


class Rwkv5FeedForward(nn.Module):
    def __init__(self, hidden, inter, layer_id):
        super().__init__()
        self.layer_id = layer_id
        self.time_mix_key = nn.Parameter(torch.rand(1, 1, hidden))
        self.time_mix_receptance = nn.Parameter(torch.rand(1, 1, hidden))
        self.key = nn.Linear(hidden, inter, bias=False)
        self.receptance = nn.Linear(hidden, hidden, bias=False)
        self.value = nn.Linear(inter, hidden, bias=False)
        self.time_shift = nn.ZeroPad2d((0, 0, 1, -1))

    def forward(self, hidden, state=None):
        if hidden.size(1) == 1 and state is not None:
            shifted = state[2][:, :, self.layer_id].unsqueeze(1)
        else:
            shifted = self.time_shift(hidden)
            if state is not None:
                shifted[:, 0] = state[2][:, :, self.layer_id]
        k = hidden * self.time_mix_key + shifted * (1 - self.time_mix_key)
        r = hidden * self.time_mix_receptance + shifted * (1 - self.time_mix_receptance)
        k = torch.square(torch.relu(self.key(k)))
        if state is not None:
            state[2][:, :, self.layer_id] = hidden[:, -1]
        return torch.sigmoid(self.receptance(r)) * self.value(k), state


class Block(nn.Module):
    def __init__(self, hidden, heads, layer_id):
        super().__init__()
        self.attention = Rwkv5SelfAttention(hidden, heads, layer_id)
        self.feed_forward = Rwkv5FeedForward(hidden, hidden * 3, layer_id)


def _state(b, c, h, layers, dev):
    s = c // h
    return [torch.zeros(b, c, layers, device=dev), torch.zeros(b, h, s, s, layers, device=dev), torch.zeros(b, c, layers, device=dev)]


@pytest.mark.parametrize("head_size", [64, 32])
def test_rwkv5_binding_matches_the_recurrence(head_size):
    from bigdl_amd.models.rwkv import bind_rwkv
    torch.manual_seed(0)
    heads, layers = 4, 2
    c = heads * head_size
    model = nn.ModuleList([Block(c, heads, i) for i in range(layers)]).cuda().float()
    x = torch.randn(2, 7, c, device="cuda") * 0.5
    x1 = torch.randn(2, 1, c, device="cuda") * 0.5
    with torch.no_grad():
        want, st_w = [], _state(2, c, heads, layers, "cuda")
        for blk in model:
            a, st_w = blk.attention(x, st_w, use_cache=True)
            f, st_w = blk.feed_forward(x, st_w)
            a1, st_w = blk.attention(x1, st_w, use_cache=True)
            f1, st_w = blk.feed_forward(x1, st_w)
            want.append((a, f, a1, f1))
        bind_rwkv(model)
        assert all(hasattr(b.attention, "_hf_forward") and hasattr(b.feed_forward, "_hf_forward") for b in model)
        st_g = _state(2, c, heads, layers, "cuda")
        for blk, (a, f, a1, f1) in zip(model, want):
            ga, st_g = blk.attention(x, st_g, use_cache=True)
            gf, st_g = blk.feed_forward(x, st_g)
            ga1, st_g = blk.attention(x1, st_g, use_cache=True)
            gf1, st_g = blk.feed_forward(x1, st_g)
            for got, ref in ((ga, a), (gf, f), (ga1, a1), (gf1, f1)):
                torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)
        for g_, w_ in zip(st_g, st_w):
            torch.testing.assert_close(g_, w_, rtol=2e-4, atol=2e-4)
