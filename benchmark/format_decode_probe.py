"""Next-token latency of the native engine per weight format (random-init Llama-2-7B, 1024-in / 32-out, fp8 KV)."""
import sys

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import LLAMA2_7B, DecoderConfig, build_random_decoder

for name in (sys.argv[1:] or ["sym_int4", "asym_int4", "woq_int4", "asym_int5", "nf4"]):
    eng = build_random_decoder(DecoderConfig(**LLAMA2_7B), Q.ggml_tensor_qtype[name], device="cuda", seed=1)
    ids = torch.randint(3, 32000, (1, 1024)).cuda()
    for _ in range(2):
        eng.generate(ids, max_new_tokens=32, kv_fp8=True)
    print(name, round(eng.rest_cost_mean * 1e3, 4), flush=True)
    del eng
    torch.cuda.empty_cache()
