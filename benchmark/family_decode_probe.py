"""Decode / first-token latency of the LayerNorm families on the native engine at their real sizes (random weights):
    PYTHONPATH=. python benchmark/family_decode_probe.py [--in-len 1024 --out-len 128 --low-bit sym_int4]
One JSON line per family."""
import argparse
import json

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models import decoder as D

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-len", type=int, default=1024)
    ap.add_argument("--out-len", type=int, default=128)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--families", default="LLAMA2_7B,BLOOM_7B,PYTHIA_6_9B,FALCON_7B,STARCODER_15B")
    args = ap.parse_args()
    qt = Q.ggml_tensor_qtype[args.low_bit]
    for name in args.families.split(","):
        cfg = D.DecoderConfig(**getattr(D, name))
        if cfg.tie_word_embeddings is False and name == "FALCON_7B":
            pass
        eng = D.build_random_decoder(cfg, qt, "cuda")
        ids = torch.randint(0, cfg.vocab_size, (1, args.in_len), device="cuda")
        for _ in range(2):
            eng.generate(ids, max_new_tokens=args.out_len)
        print(json.dumps({"family": name, "low_bit": args.low_bit, "in_out": f"{args.in_len}-{args.out_len}",
                          "first_token_ms": round(eng.first_cost * 1e3, 2), "next_token_ms": round(eng.rest_cost_mean * 1e3, 3),
                          "weights_gb": round(eng.weight_bytes() / 1e9, 2), "peak_gb": round(eng.peak_memory, 2)}), flush=True)
        del eng
        torch.cuda.empty_cache()
