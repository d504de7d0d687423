"""Diffusion pipelines (Stable Diffusion 1.x / 2.x / XL, LCM-LoRA) and sentence-transformers wrappers.

The reference swaps the UNet's attention processor for one that calls its fused attention kernel, zero-padding the
40-channel heads of SD 1.x to 64 (python/llm/src/ipex_llm/transformers/models/sd.py:45-142), and keeps the SDXL VAE
decoder from overflowing in fp16 (``upcast_vae``, :145-152); both are installed by ``optimize_model(pipe, low_bit=None)``
(transformers/convert.py:1251-1263).  A ``SentenceTransformer`` wrapper gets the BERT forwards of its first module
(convert.py:1265-1281).

Here the processor hands q / k / v to the MFMA flash-attention kernel (``csrc/hip/attn_prefill.hip``, bidirectional):
head sizes 64 / 80 / 96 / 128 / 256 as they are, 40 zero-padded to 64 (the padded channels add nothing to q . k and
produce zero output channels that are cut off again), everything else (160-channel heads of the deepest SD 1.x blocks,
masks, fp32 or CPU tensors, training) through PyTorch's scaled_dot_product_attention.  ``diffusers`` is not imported:
the processor only relies on the attribute protocol of ``diffusers.models.attention_processor.Attention``."""
import math
from types import MethodType

import torch
import torch.nn.functional as F

_PAD_TO = {40: 64}                       # head size -> kernel head size
_KERNEL_HEAD_DIMS = (64, 72, 80, 96, 128, 256)


def fused_bidirectional_attention(query, key, value, scale):
    """q [B, H, Sq, D], k / v [B, H, Lk, D] -> [B, H, Sq, D] through the flash kernel, or None when it does not apply."""
    d = query.shape[-1]
    dk = _PAD_TO.get(d, d)
    if query.device.type != "cuda" or query.dtype != torch.float16 or torch.is_grad_enabled() or dk not in _KERNEL_HEAD_DIMS:
        return None
    from ..ops import hip
    if dk != d:
        query, key, value = (F.pad(t, (0, dk - d)) for t in (query, key, value))
    fix = lambda t: t if (t.stride(-1) == 1 and all(t.stride(i) % 8 == 0 for i in (0, 1, 2))) else t.contiguous()  # noqa: E731
    b, h, sq, _ = query.shape
    out = hip.attn_prefill(fix(query), fix(key), fix(value), scale=scale, causal=False)       # [B, Sq, H * dk]
    return out.view(b, sq, h, dk)[..., :d].transpose(1, 2)


class HipAttnProcessor:
    """Drop-in for diffusers' ``AttnProcessor2_0`` (self- and cross-attention of UNet / VAE blocks)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        residual = hidden_states
        if getattr(attn, "spatial_norm", None) is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            batch, channel, height, width = hidden_states.shape
            hidden_states = hidden_states.view(batch, channel, height * width).transpose(1, 2)
        batch, kv_len, _ = hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape
        heads = attn.heads
        if attention_mask is not None:
            attention_mask = attn.prepare_attention_mask(attention_mask, kv_len, batch)
            attention_mask = attention_mask.view(batch, heads, -1, attention_mask.shape[-1])
        if getattr(attn, "group_norm", None) is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        elif getattr(attn, "norm_cross", None):
            encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        head_dim = key.shape[-1] // heads
        query = query.view(batch, -1, heads, head_dim).transpose(1, 2)
        key = key.view(batch, -1, heads, head_dim).transpose(1, 2)
        value = value.view(batch, -1, heads, head_dim).transpose(1, 2)
        if getattr(attn, "norm_q", None) is not None:
            query = attn.norm_q(query)
        if getattr(attn, "norm_k", None) is not None:
            key = attn.norm_k(key)
        scale = getattr(attn, "scale", None) or 1.0 / math.sqrt(head_dim)
        out = fused_bidirectional_attention(query, key, value, scale) if attention_mask is None else None
        if out is None:
            out = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0,
                                                 is_causal=False, scale=scale)
        hidden_states = out.transpose(1, 2).reshape(batch, -1, heads * head_dim).to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(batch, channel, height, width)
        if getattr(attn, "residual_connection", False):
            hidden_states = hidden_states + residual
        return hidden_states / getattr(attn, "rescale_output_factor", 1.0)


def upcast_vae(self):
    """SDXL's VAE decoder overflows fp16 in its up blocks: run those, the output norm and the output convolution in
    bf16 (same exponent range as fp32, MFMA rate of fp16) instead of moving the whole VAE to fp32 as diffusers does."""
    dec = self.vae.decoder
    dec.up_blocks.to(torch.bfloat16)
    dec.conv_norm_out.to(torch.bfloat16)
    dec.conv_out.to(torch.bfloat16)


def is_diffusion_pipeline(obj) -> bool:
    """A diffusers pipeline, recognised without importing diffusers: not a module itself, carries a denoiser
    (``unet`` or ``transformer``) that takes attention processors."""
    if isinstance(obj, torch.nn.Module):
        return False
    den = getattr(obj, "unet", None) or getattr(obj, "transformer", None)
    return den is not None and callable(getattr(den, "set_attn_processor", None))


def optimize_diffusion_pipeline(pipe, low_bit=None, **kwargs):
    """``optimize_model(pipe, low_bit=None)`` for a diffusion pipeline: the fused attention processor on the denoiser
    (and on the VAE when it takes processors), the bf16 decoder guard on SDXL pipelines; with ``low_bit`` the
    denoiser's linears are quantized as well (it has to be on the CPU for that, like any model)."""
    den = getattr(pipe, "unet", None) or getattr(pipe, "transformer", None)
    if low_bit not in (None, "fp16", "bf16"):
        from ..optimize import optimize_model
        optimize_model(den, low_bit=low_bit, optimize_llm=False, **kwargs)
    den.set_attn_processor(HipAttnProcessor())
    vae = getattr(pipe, "vae", None)
    if vae is not None and callable(getattr(vae, "set_attn_processor", None)):
        vae.set_attn_processor(HipAttnProcessor())
    if "XL" in pipe.__class__.__name__ and vae is not None:
        pipe.upcast_vae = MethodType(upcast_vae, pipe)
    return pipe


def unwrap_sentence_transformer(model):
    """The HuggingFace model inside a ``SentenceTransformer`` (its first module's ``auto_model``), or None."""
    if model.__class__.__name__ != "SentenceTransformer":
        return None
    first = getattr(model, "_modules", {}).get("0")
    return getattr(first, "auto_model", None)
