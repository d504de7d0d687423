"""The remote-code forwards (bigdl_amd/models/remote_code.py) on the GPU: fp16 tensors go through the MFMA flash kernel,
results must match the stand-ins' fp32 statements (tests/test_remote_code_cpu.py) on the host."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def standins():
    spec = importlib.util.spec_from_file_location("remote_code_standins",
                                                  os.path.join(os.path.dirname(__file__), "test_remote_code_cpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.H, mod.NH, mod.D = 256, 4, 64            # a head size the flash kernel takes
    return mod


def _rel(a, b):
    return ((a.float().cpu() - b).abs().mean() / b.abs().mean()).item()


def test_baichuan_m1_and_chatglm_v1_on_the_flash_kernel(standins):
    from bigdl_amd.models import remote_code as rc
    from bigdl_amd.ops import hip
    from bigdl_amd.transformers.kv import DynamicNormalCache
    torch.manual_seed(0)
    calls = []
    real = hip.attn_prefill
    hip.attn_prefill = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        att = standins.BaichuanAttention().eval()
        x = torch.randn(2, 40, 256)
        with torch.no_grad():
            want = att.reference(x)
            rc.bind_remote_code_forwards(att)
            att = att.half().cuda()
            att.rotary_emb.inv_freq = att.rotary_emb.inv_freq.cuda()
            cache = DynamicNormalCache()
            got, _, _ = att(x[:, :32].half().cuda(), past_key_value=cache, use_cache=True)
            assert _rel(got, want[:, :32]) < 0.02
            for t in range(32, 40):
                step, _, _ = att(x[:, t:t + 1].half().cuda(), past_key_value=cache, use_cache=True,
                                 position_ids=torch.tensor([[t]], device="cuda").expand(2, 1))
                assert _rel(step[:, 0], want[:, t]) < 0.03
        n_b = len(calls)
        assert n_b >= 1
        glm = standins.SelfAttention(True).eval()
        s, b = 24, 2
        x = torch.randn(s, b, 256)
        pos = torch.cat([torch.arange(16), torch.full((8,), 15)])
        blk = torch.cat([torch.zeros(16, dtype=torch.long), torch.arange(1, 9)])
        position_ids = torch.stack([pos, blk])[None].expand(b, 2, s)
        with torch.no_grad():
            want = glm.reference(x, position_ids)
            rc.bind_remote_code_forwards(glm)
            glm = glm.half().cuda()
            none = torch.zeros(1, 1, 1, 1, dtype=torch.bool, device="cuda")
            out, present = glm(x[:16].half().cuda(), position_ids[..., :16].cuda(), none, 0, use_cache=True)
            assert _rel(out, want[:16]) < 0.02
            out, present = glm(x[16:17].half().cuda(), position_ids[..., 16:17].cuda(), none, 0, layer_past=present, use_cache=True)
            assert _rel(out[0], want[16]) < 0.03
        assert len(calls) > n_b
    finally:
        hip.attn_prefill = real


def test_phixtral_experts_low_bit_on_gpu(standins):
    from bigdl_amd.optimize import optimize_model
    torch.manual_seed(1)
    moe = standins.MoE().eval()
    x = torch.randn(1, 12, 256)
    with torch.no_grad():
        want = moe(x)
    q = optimize_model(standins._Holder(moe, "phi-msft"), low_bit="sym_int8").half().cuda()
    with torch.no_grad():
        got = q.inner(x.half().cuda())
        one = q.inner(x[:, :1].half().cuda())
    assert _rel(got, want) < 0.03 and _rel(one, want[:, :1]) < 0.03
