"""Decode step time of the native engine at a given batch size and context length (random-init Llama-2-7B sym_int4,
fp8 KV): prints one JSON line per batch size.  The cache is filled by a real prompt pass of ``--ctx`` tokens per row.
    PYTHONPATH=. python benchmark/batch_decode_probe.py --batch 1,8,32,64 --ctx 1024 [--steps 32]
Run under ``rocprofv3 --kernel-trace --stats`` with one batch size to attribute the step to kernels."""
import argparse
import json
import time

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import LLAMA2_7B, DecoderConfig, build_random_decoder

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", default="1")
    ap.add_argument("--ctx", type=int, default=32)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--kv", default="fp8", choices=["fp8", "fp16", "int4"])
    ap.add_argument("--model", default="LLAMA2_7B", help="a shape constant of bigdl_amd.models.decoder (LLAMA3_8B, TINY_LLAMA, QWEN2_7B ...)")
    args = ap.parse_args()
    import bigdl_amd.models.decoder as _D
    shape = DecoderConfig(**getattr(_D, args.model))
    eng = build_random_decoder(shape, Q.SYM_INT4, device="cuda", seed=1)
    for b in [int(x) for x in args.batch.split(",")]:
        eng.ensure_cache(b, args.ctx + args.steps + 8, fp8={"fp8": True, "fp16": False, "int4": "int4"}[args.kv])
        eng.cache.reset()
        ids = torch.randint(3, shape.vocab_size, (b, args.ctx)).cuda()
        logits = eng.prefill(ids)
        tok = logits[:, -1].argmax(-1).view(b, 1)
        for _ in range(3):                             # graph capture + warm-up
            _, nxt = eng.decode(tok)
            tok = nxt.view(b, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, nxt = eng.decode(tok)
            tok = nxt.view(b, 1)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        kv_gb = {"fp8": 1, "fp16": 2, "int4": 0.5625}[args.kv] * 2 * b * (args.ctx + args.steps // 2) * shape.kv_size * shape.num_layers / 1e9
        print(json.dumps({"kv": args.kv, "batch": b, "ctx": args.ctx, "ms_per_step": round(ms, 3), "tokens_per_s": round(b / ms * 1e3, 1),
                          "kv_read_gb": round(kv_gb, 2), "weights_gb": round(eng.weight_bytes() / 1e9, 2),
                          "hbm_floor_ms": round((kv_gb + eng.weight_bytes() / 1e9) / 8e3 * 1e3, 3)}), flush=True)
