"""Headline benchmark: INT4 (sym_int4) Llama-2-7B, 1024-token prompt / 128 new tokens, batch 1, greedy —
the reference's all-in-one protocol (python/llm/dev/benchmark/all-in-one/run.py:63-106, 465-569; metric
definitions utils/benchmark_util_4_47.py:3372-3393): "1st token latency" = wall time of the prefill forward
+ first sample incl. device sync, "2+ avg latency" = mean wall time of every later token.

    python bench.py --gpus N --steps K --warmup W

A step is one full generation trial (prefill + 127 decode steps).  W untimed warm-up trials, then exactly K
timed trials bracketed by barrier + synchronize; the max over ranks is reported by rank 0 as one JSON line.
Weights are random-init sym_int4 blocks of the Llama-2-7B architecture, prompts are synthetic token ids
(no network for checkpoints/datasets).  With N > 1 every GPU serves its own batch-1 stream (weak scaling);
`value` stays the per-stream next-token latency and `aggregate_tokens_per_s` sums the streams.  `--parallelism tp`
shards one stream over the N GPUs instead (strong scaling of the same latency metric).
"""
import argparse
import json
import os
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--in-len", type=int, default=1024)
    ap.add_argument("--out-len", type=int, default=128)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--kv", default="fp8", choices=["fp8", "fp16", "int4"])
    ap.add_argument("--model", default="llama2-7b", choices=["llama2-7b", "tinyllama"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--parallelism", default="dp", choices=["dp", "tp"],
                    help="N > 1: dp = every GPU serves its own batch-1 stream (default); tp = one stream, weights "
                         "sharded over the GPUs (heads / MLP columns), all-reduce after o_proj and down_proj")
    ap.add_argument("--rccl-allreduce", action="store_true", help="tp: RCCL all-reduce instead of the one-shot IPC kernel")
    ap.add_argument("--prefill-dense-cache", action="store_true",
                    help="keep fp16 copies of the weights for prefill GEMMs (off in the headline number)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, LLAMA2_7B, TINY_LLAMA, build_random_decoder

    cfg = DecoderConfig(**(LLAMA2_7B if args.model == "llama2-7b" else TINY_LLAMA))
    qt = Q.ggml_tensor_qtype[args.low_bit]
    model = build_random_decoder(cfg, qt, device=f"cuda:{local_rank}", seed=1234)
    tp = dist_on and args.parallelism == "tp"
    if tp:
        from bigdl_amd.parallel.tensor_parallel import shard_engine
        model = shard_engine(model, rank, world, fast_allreduce=not args.rccl_allreduce)
    model.use_graph = not args.no_graph
    if args.prefill_dense_cache:
        from bigdl_amd.models.decoder import Lin
        Lin.cache_dense = True
    g = torch.Generator().manual_seed(0 if tp else rank)
    prompt = torch.randint(3, cfg.vocab_size, (1, args.in_len), generator=g).cuda()

    def trial():
        out = model.generate(prompt, max_new_tokens=args.out_len, min_new_tokens=args.out_len, do_sample=False,
                             kv_fp8={"fp8": True, "fp16": False, "int4": "int4"}[args.kv])
        assert out.shape[1] == args.in_len + args.out_len
        return model.first_cost, model.rest_cost_mean, model.peak_memory

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trial()
    barrier()
    t0 = time.perf_counter()
    res = [trial() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0

    first = sum(r[0] for r in res) / len(res) * 1e3
    rest = sum(r[1] for r in res) / len(res) * 1e3
    peak = max(r[2] for r in res)
    stats = torch.tensor([elapsed, first, rest, peak], dtype=torch.float64, device="cuda")
    if dist_on:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    elapsed, first, rest, peak = stats.tolist()
    if rank == 0:
        print(json.dumps({
            "metric": "next-token latency (ms/token), INT4 Llama-2-7B, 1024-in/128-out, batch 1 greedy",
            "value": round(rest, 4), "unit": "ms/token", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": False, "scaling": "strong" if tp else "weak",
            "vs_baseline": None, "dtype": "fp16 activations, sym_int4 weights, %s KV" % args.kv, "data": "synthetic",
            "first_token_latency_ms": round(first, 3), "next_token_latency_ms": round(rest, 4),
            "peak_mem_gb": round(peak, 3), "aggregate_tokens_per_s": round((1 if tp else world) * 1000.0 / rest, 1),
            "weight_read_floor_ms": round(model.weight_bytes() / 8e12 * 1e3, 3),
            "config": {"model": "Llama-2-7B (random-init)" if args.model == "llama2-7b" else "TinyLlama-1.1B (random-init)",
                       "low_bit": args.low_bit, "global_batch": 1 if tp else world, "seq_len": args.in_len,
                       "new_tokens": args.out_len,
                       "parallelism": (f"tp{world}" if tp else f"dp{world}") if world > 1 else "single",
                       "hip_graph": not args.no_graph, "prefill_dense_cache": bool(args.prefill_dense_cache)}}))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
