"""Decode step time of the native engine at a given batch size (random-init Llama-2-7B sym_int4, 32-token prompts,
32 new tokens, fp8 KV): prints `<batch> <ms per step>`.  Used to attribute how the step scales with the row count."""
import sys

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import LLAMA2_7B, DecoderConfig, build_random_decoder

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = build_random_decoder(DecoderConfig(**LLAMA2_7B), Q.SYM_INT4, device="cuda", seed=1)
ids = torch.randint(3, 32000, (b, 32)).cuda()
for _ in range(2):
    eng.generate(ids, max_new_tokens=32, kv_fp8=True)
print(b, round(eng.rest_cost_mean * 1e3, 4))
