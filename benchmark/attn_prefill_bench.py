"""Time of the prefill attention kernel alone (Llama-2-7B shape by default) vs PyTorch SDPA."""
import json
import sys

import torch
import torch.nn.functional as F

from bigdl_amd.ops import hip

b, h, hkv, s, d = 1, 32, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 128
q, k = (torch.randn(b, x, s, d, device="cuda").half() for x in (h, hkv))
vt = torch.randn(b, hkv, d, s, device="cuda").half()
v = vt.transpose(-1, -2).contiguous()


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


flops = 4 * s * s * d * h * b / 2
for name, fn in (("hip_vt", lambda: hip.attn_prefill(q, k, vt, scale=d ** -0.5, causal=True, v_transposed=True)),
                 ("hip_v", lambda: hip.attn_prefill(q, k, v, scale=d ** -0.5, causal=True)),
                 ("torch_sdpa", lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))):
    us = timeit(fn)
    print(json.dumps({"kernel": name, "S": s, "us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1)}), flush=True)
