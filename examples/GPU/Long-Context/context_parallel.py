"""First-token latency of a very long prompt divided over the GPUs of a node: context-parallel prefill (zigzag
chunks, K / V all-gathered per layer over RCCL), then replicated decode from the complete cache on every rank.

    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 context_parallel.py --prompt-tokens 131072

Random-init Llama-2-7B-shaped weights (sym_int4), synthetic prompt; --model-path loads a real checkpoint instead.
"""
import argparse
import os
import time

import torch
import torch.distributed as dist

from bigdl_amd import qtypes as Q
from bigdl_amd.models import decoder as D

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt-tokens", type=int, default=32768)
    ap.add_argument("--new-tokens", type=int, default=32)
    ap.add_argument("--model-path", default=None)
    ap.add_argument("--low-bit", default="sym_int4")
    args = ap.parse_args()
    rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(rank)
    if "RANK" in os.environ:
        dist.init_process_group("nccl")
    world = dist.get_world_size() if dist.is_initialized() else 1
    if args.model_path:
        from ipex_llm.transformers import AutoModelForCausalLM
        from bigdl_amd.models.hf_patch import get_native_engine
        model = AutoModelForCausalLM.from_pretrained(args.model_path, load_in_low_bit=args.low_bit).half().to("cuda")
        eng = get_native_engine(model)
    else:
        cfg = D.DecoderConfig(**{**D.LLAMA2_7B, "max_position": args.prompt_tokens + args.new_tokens})
        eng = D.build_random_decoder(cfg, Q.ggml_tensor_qtype[args.low_bit], "cuda")
    s = args.prompt_tokens // (2 * world) * (2 * world)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, eng.cfg.vocab_size, (1, s), generator=g).cuda()
    eng.ensure_cache(1, s + args.new_tokens + 1, fp8=True)
    for it in range(2):                                  # first pass warms the allocator and the GEMM heuristics
        eng.cache.reset()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        logits = eng.prefill_context_parallel(ids)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
    tok = logits[:, -1].argmax(-1).view(1, 1)
    _, nxt = eng.decode(tok)                             # untimed: captures the decode graph
    tok = nxt.view(1, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.new_tokens):
        _, nxt = eng.decode(tok)
        tok = nxt.view(1, 1)
    torch.cuda.synchronize()
    rest = (time.perf_counter() - t0) / args.new_tokens
    if rank == 0:
        print(f"world {world}  prompt {s} tokens  first token {first * 1e3:.1f} ms  next token {rest * 1e3:.2f} ms")
    if dist.is_initialized():
        dist.destroy_process_group()
