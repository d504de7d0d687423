"""Diffusion-pipeline and sentence-transformers entry points of ``optimize_model`` (reference
transformers/models/sd.py, convert.py:1251-1281).  ``diffusers`` and ``sentence_transformers`` are not installed:
the attention block below follows the attribute protocol of ``diffusers.models.attention_processor.Attention`` and the
wrapper that of ``SentenceTransformer`` (parity with the real packages unpinned)."""
import math

import pytest
import torch
import torch.nn as nn

from bigdl_amd.models.sd import HipAttnProcessor, fused_bidirectional_attention
from bigdl_amd.optimize import optimize_model


class Attention(nn.Module):
    def __init__(self, query_dim=64, cross_dim=None, heads=2, dim_head=40, residual=False, group_norm=False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.spatial_norm = self.norm_q = self.norm_k = None
        self.group_norm = nn.GroupNorm(4, query_dim) if group_norm else None
        self.norm_cross = False
        self.residual_connection, self.rescale_output_factor = residual, 2.0 if residual else 1.0
        self.processor = None

    def prepare_attention_mask(self, mask, target_length, batch_size):
        return mask.repeat_interleave(self.heads, dim=0)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask)


def reference(attn, x, ctx=None, mask=None):
    res = x
    if x.ndim == 4:
        b, c, h, w = x.shape
        x = x.view(b, c, h * w).transpose(1, 2)
    if attn.group_norm is not None:
        x = attn.group_norm(x.transpose(1, 2)).transpose(1, 2)
    ctx = x if ctx is None else ctx
    b = x.shape[0]
    sp = lambda t: t.view(b, -1, attn.heads, t.shape[-1] // attn.heads).transpose(1, 2)      # noqa: E731
    q, k, v = sp(attn.to_q(x)), sp(attn.to_k(ctx)), sp(attn.to_v(ctx))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if mask is not None:
        s = s + mask[:, None]
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(b, -1, attn.heads * q.shape[-1])
    o = attn.to_out[0](o)
    if res.ndim == 4:
        o = o.transpose(-1, -2).reshape(res.shape)
    if attn.residual_connection:
        o = o + res
    return o / attn.rescale_output_factor


class Unet(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn = Attention()
        self.cross_attn = Attention(cross_dim=48)
        self.vae_like = Attention(residual=True, group_norm=True, heads=1, dim_head=64)

    def set_attn_processor(self, proc):
        for m in self.modules():
            if isinstance(m, Attention):
                m.processor = proc


class StableDiffusionXLPipeline:
    def __init__(self):
        self.unet = Unet()
        self.vae = type("V", (), {})()
        self.vae.decoder = nn.ModuleDict(dict(up_blocks=nn.ModuleList([nn.Conv2d(4, 4, 1)]), conv_norm_out=nn.GroupNorm(2, 4),
                                              conv_out=nn.Conv2d(4, 3, 1)))
        self.vae.decoder.up_blocks, self.vae.decoder.conv_norm_out, self.vae.decoder.conv_out = (
            self.vae.decoder["up_blocks"], self.vae.decoder["conv_norm_out"], self.vae.decoder["conv_out"])


def test_pipeline_gets_the_processor_and_the_vae_guard():
    torch.manual_seed(0)
    pipe = optimize_model(StableDiffusionXLPipeline(), low_bit=None)
    u = pipe.unet
    assert all(isinstance(m.processor, HipAttnProcessor) for m in u.modules() if isinstance(m, Attention))
    x, ctx = torch.randn(2, 10, 64), torch.randn(2, 7, 48)
    mask = torch.zeros(2, 1, 7)
    mask[:, :, 5:] = -1e4
    img = torch.randn(2, 64, 3, 4)
    with torch.no_grad():
        assert torch.allclose(u.self_attn(x), reference(u.self_attn, x), atol=1e-5)
        assert torch.allclose(u.cross_attn(x, ctx), reference(u.cross_attn, x, ctx), atol=1e-5)
        assert torch.allclose(u.cross_attn(x, ctx, mask), reference(u.cross_attn, x, ctx, mask), atol=1e-5)
        assert torch.allclose(u.vae_like(img), reference(u.vae_like, img), atol=1e-5)
    pipe.upcast_vae()
    assert pipe.vae.decoder.conv_out.weight.dtype == torch.bfloat16
    assert fused_bidirectional_attention(torch.randn(1, 2, 4, 40), torch.randn(1, 2, 4, 40), torch.randn(1, 2, 4, 40), 0.1) is None


def test_pipeline_denoiser_can_be_quantized():
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear
    torch.manual_seed(0)
    pipe = StableDiffusionXLPipeline()
    x = torch.randn(2, 10, 64)
    with torch.no_grad():
        pipe.unet.set_attn_processor(HipAttnProcessor())
        ref = pipe.unet.self_attn(x)
        pipe = optimize_model(pipe, low_bit="sym_int8")
        assert isinstance(pipe.unet.self_attn.to_q, LowBitLinear)
        got = pipe.unet.self_attn(x)
    assert (got - ref).abs().max() / ref.abs().max() < 0.05


def test_sentence_transformer_wrapper_converts_its_encoder():
    from transformers import BertConfig, BertModel
    from bigdl_amd.transformers.low_bit_linear import LowBitLinear

    class Transformer(nn.Module):
        def __init__(self):
            super().__init__()
            self.auto_model = BertModel(BertConfig(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                                                   intermediate_size=128, max_position_embeddings=64)).eval()

        def forward(self, ids):
            return self.auto_model(ids).last_hidden_state

    class SentenceTransformer(nn.Sequential):
        pass

    torch.manual_seed(0)
    st = SentenceTransformer(Transformer())
    ids = torch.randint(0, 100, (2, 12))
    with torch.no_grad():
        ref = st(ids)
        out = optimize_model(st, low_bit="sym_int8")
        assert out is st and any(isinstance(m, LowBitLinear) for m in st.modules())
        got = st(ids)
    assert (got - ref).abs().max() / ref.abs().max() < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("dim_head,heads", [(40, 8), (64, 10), (80, 8), (160, 8)])
def test_processor_on_the_flash_kernel(dim_head, heads):
    """fp16 on the GPU: 40-channel heads zero-padded to 64, 64 / 80 as they are, 160 through PyTorch; against fp32."""
    torch.manual_seed(0)
    a = Attention(query_dim=320, cross_dim=768, heads=heads, dim_head=dim_head)
    s = Attention(query_dim=320, heads=heads, dim_head=dim_head)
    for m in (a, s):
        m.processor = HipAttnProcessor()
    x, ctx = torch.randn(2, 1024, 320), torch.randn(2, 77, 768)
    with torch.no_grad():
        ref_c, ref_s = reference(a, x, ctx), reference(s, x)
        a, s = a.half().cuda(), s.half().cuda()
        got_c = a(x.half().cuda(), ctx.half().cuda()).float().cpu()
        got_s = s(x.half().cuda()).float().cpu()
    assert (got_c - ref_c).abs().max() / ref_c.abs().max() < 2e-2
    assert (got_s - ref_s).abs().max() / ref_s.abs().max() < 2e-2
    if dim_head != 160:
        q = torch.randn(1, heads, 64, dim_head, device="cuda", dtype=torch.float16)
        with torch.no_grad():
            assert fused_bidirectional_attention(q, q, q, dim_head ** -0.5) is not None     # (the kernel ran, not the fallback)
