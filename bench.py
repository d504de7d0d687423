"""Headline benchmark: INT4 (sym_int4) Llama-2-7B, 1024-token prompt / 128 new tokens, batch 1, greedy —
the reference's all-in-one protocol (python/llm/dev/benchmark/all-in-one/run.py:63-106, 465-569; metric
definitions utils/benchmark_util_4_47.py:3372-3393): "1st token latency" = wall time of the prefill forward
+ first sample incl. device sync, "2+ avg latency" = mean wall time of every later token.

    python bench.py --gpus N --steps K --warmup W [--model M] [--parallelism dp|tp|pp|ep] [--bench decode|qlora]

A step is one full generation trial (prefill + 127 decode steps).  W untimed warm-up trials, then exactly K
timed trials bracketed by barrier + synchronize; the max over ranks is reported by rank 0 as one JSON line.
Weights are random-init low-bit blocks of the named architecture, prompts are synthetic token ids (no network for
checkpoints / datasets).

``--gpus N`` with N > 1 and no launcher in the environment (no WORLD_SIZE): this process starts N ranks itself, one per
GPU, before it touches a device, and exits with their status; under ``torch.distributed.run`` the ranks come from the
environment as usual.  ``n_gpus`` in the output is the size of the process group that actually ran.

Parallelism of the decode benchmark (reference protocol for the multi-GPU modes: run.py:1954-2060 pipeline parallel,
2207-2284 DeepSpeed AutoTP):
  dp  every GPU serves its own batch-1 stream (weak scaling; ``value`` = per-stream latency, tokens/s summed)
  tp  one stream, heads / MLP columns sharded, all-reduce after o_proj and down_proj (strong scaling)
  pp  one stream, consecutive layers per GPU, activations handed from stage to stage (strong scaling)
  ep  mixture-of-experts models: experts sharded, attention replicated, every GPU its own stream (weak scaling)
``--bench qlora`` times QLoRA training steps instead (benchmark/qlora_bench.py; data parallel over the GPUs).

``BIGDL_FORCE_CPU=1`` runs the same driver on the host over gloo with the HuggingFace module path (test-size model
``--model tiny``): the plumbing check of the launcher and of the parallel modes where there is no GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

MODELS = ("llama2-7b", "tinyllama", "llama3-8b", "llama3-70b", "mixtral-8x7b", "tiny")
LLAMA3_70B = dict(vocab_size=128256, hidden_size=8192, intermediate_size=28672, num_layers=80, num_heads=64,
                  num_kv_heads=8, head_dim=128, rope_theta=500000.0, max_position=8192)
TINY = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_layers=4, num_heads=4, num_kv_heads=2, head_dim=64,
            max_position=2048)
TINY_MOE = dict(TINY, num_experts=4, top_k=2, moe_intermediate_size=256, model_type="mixtral")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--in-len", type=int, default=1024)
    ap.add_argument("--out-len", type=int, default=128)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--kv", default="fp8", choices=["fp8", "fp16", "int4"])
    ap.add_argument("--model", default="llama2-7b", choices=MODELS)
    ap.add_argument("--bench", default="decode", choices=["decode", "qlora"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--parallelism", default="dp", choices=["dp", "tp", "pp", "ep"])
    ap.add_argument("--rccl-allreduce", action="store_true", help="tp: RCCL all-reduce instead of the one-shot IPC kernel")
    ap.add_argument("--prefill-dense-cache", action="store_true",
                    help="keep fp16 copies of the weights for prefill GEMMs (off in the headline number)")
    ap.add_argument("--prefill-fp8", action="store_true",
                    help="prompt GEMMs on hipBLASLt's fp8 path, activations and weights quantized per row to e4m3: lower "
                         "precision than the headline configuration, reported as such (off in the headline number)")
    return ap.parse_args()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """Start ``n`` copies of this script, one rank each (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), and wait.  Called
    before anything in this process has touched a GPU."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def model_config(name, moe_ok=True):
    from bigdl_amd.models.decoder import DecoderConfig, LLAMA2_7B, LLAMA3_8B, MIXTRAL_8X7B, TINY_LLAMA
    table = {"llama2-7b": (LLAMA2_7B, "Llama-2-7B"), "tinyllama": (TINY_LLAMA, "TinyLlama-1.1B"),
             "llama3-8b": (LLAMA3_8B, "Llama-3-8B"), "llama3-70b": (LLAMA3_70B, "Llama-3-70B"),
             "mixtral-8x7b": (MIXTRAL_8X7B, "Mixtral-8x7B"), "tiny": (TINY, "test-size llama")}
    d, label = table[name]
    return DecoderConfig(**d), label + " (random-init)"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = os.environ.get("BIGDL_FORCE_CPU") == "1"
    dist_on = world > 1
    dist = None
    if not cpu:
        torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        if cpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()          # what the process group really spans
    if args.bench == "qlora":
        return qlora(args, rank, world, dist)
    runner = cpu_module_path if cpu else native_engine
    runner(args, rank, local_rank, world, dist)
    if dist_on:
        dist.barrier()                         # nobody tears its connections down while a peer still reads from them
        dist.destroy_process_group()


def report(args, world, par, label, stats, floor_ms=None, extra=None):
    elapsed, first, rest, peak = stats
    strong = par in ("tp", "pp") and world > 1
    streams = 1 if strong else world
    out = {
        "metric": f"next-token latency (ms/token), {'INT4' if args.low_bit == 'sym_int4' else args.low_bit} "
                  f"{label.split(' (')[0]}, {args.in_len}-in/{args.out_len}-out, batch 1 greedy",
        "value": round(rest, 4), "unit": "ms/token", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": False,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": f"fp16 activations, {args.low_bit} weights, {args.kv} KV" + (", e4m3 x e4m3 prompt GEMMs (reduced precision)"
                                                                              if getattr(args, "prefill_fp8", False) else ""),
        "data": "synthetic",
        "first_token_latency_ms": round(first, 3), "next_token_latency_ms": round(rest, 4), "peak_mem_gb": round(peak, 3),
        "aggregate_tokens_per_s": round(streams * 1000.0 / rest, 1),
        "config": {"model": label, "low_bit": args.low_bit, "global_batch": streams, "seq_len": args.in_len,
                   "new_tokens": args.out_len, "parallelism": f"{par}{world}" if world > 1 else "single",
                   "hip_graph": not args.no_graph, "prefill_dense_cache": bool(args.prefill_dense_cache),
                   "prefill_fp8": bool(getattr(args, "prefill_fp8", False))}}
    if floor_ms is not None:
        out["weight_read_floor_ms"] = round(floor_ms, 3)
    out.update(extra or {})
    print(json.dumps(out), flush=True)


def timed_trials(args, trial, dist, device):
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()
        if device.type == "cuda":
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
    stats = torch.tensor([elapsed, first, rest, peak], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    return stats.tolist()


def native_engine(args, rank, local_rank, world, dist):
    """the decode benchmark on the native engine (one MI355X per rank)"""
    import dataclasses
    import torch
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import build_random_decoder
    cfg, label = model_config(args.model)
    qt = Q.ggml_tensor_qtype[args.low_bit]
    par = args.parallelism if world > 1 else "dp"
    dev = torch.device("cuda", local_rank)
    full_bytes = None
    if par == "tp":
        # every rank builds its own shard directly (a 70B model is never materialised whole): 1/world of the heads and
        # of the MLP columns, kv heads replicated when the ranks outnumber them, embeddings and lm_head replicated
        assert cfg.num_heads % world == 0 and cfg.intermediate_size % world == 0
        shard = dataclasses.replace(cfg, num_heads=cfg.num_heads // world, num_kv_heads=max(cfg.num_kv_heads // world, 1),
                                    intermediate_size=cfg.intermediate_size // world)
        model = build_random_decoder(shard, qt, device=str(dev), seed=1234 + rank)
        model.tp_group, model.tp_world = dist.group.WORLD, world
        if not args.rccl_allreduce:
            from bigdl_amd.parallel.xgmi_allreduce import XgmiAllReduce
            model.tp_fast_allreduce = XgmiAllReduce.create_verified(model.tp_group, device=dev)   # None: RCCL
    elif par == "pp":
        from bigdl_amd.parallel.pipeline import EnginePipeline, build_random_stage
        model = EnginePipeline(build_random_stage(cfg, qt, rank, world, device=str(dev), seed=1234))
    else:
        model = build_random_decoder(cfg, qt, device=str(dev), seed=1234)
        if par == "ep":
            from bigdl_amd.parallel.expert_parallel import shard_engine_experts
            full_bytes = model.weight_bytes()
            shard_engine_experts(model, rank, world, dist.group.WORLD, mode="a2a")
    model.use_graph = not args.no_graph
    if args.prefill_dense_cache:
        from bigdl_amd.models.decoder import Lin
        Lin.cache_dense = True
    if args.prefill_fp8:
        model.prefill_fp8 = True
    g = torch.Generator().manual_seed(0 if par in ("tp", "pp") else rank)
    prompt = torch.randint(3, cfg.vocab_size, (1, args.in_len), generator=g).to(dev)

    def trial():
        out = model.generate(prompt, max_new_tokens=args.out_len, min_new_tokens=args.out_len, do_sample=False,
                             kv_fp8={"fp8": True, "fp16": False, "int4": "int4"}[args.kv])
        assert out.shape[1] == args.in_len + args.out_len
        return model.first_cost, model.rest_cost_mean, model.peak_memory

    stats = timed_trials(args, trial, dist, dev)
    if os.environ.get("BIGDL_PMC_CALIBRATE") == "1":
        # a known amount of read traffic for benchmark/pmc_summary.py: three reductions over exactly 1 GiB
        ref = torch.zeros(1 << 28, dtype=torch.float32, device=dev)
        for _ in range(3):
            ref.sum()
        torch.cuda.synchronize()
    if rank == 0:
        # bytes a decoded token must stream (routed experts: only the activated ones) over the HBM peak
        read = model.decode_read_bytes() if hasattr(model, "decode_read_bytes") else model.weight_bytes()
        report(args, world, par, label, stats, floor_ms=read / 8e12 * 1e3)


def cpu_module_path(args, rank, local_rank, world, dist):
    """BIGDL_FORCE_CPU=1: the same driver on the host (gloo) with the HuggingFace module path"""
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM
    from bigdl_amd.optimize import optimize_model
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(world, 1)))
    d = TINY if args.model == "tiny" else dict(model_config(args.model)[0].__dict__)
    cfg = LlamaConfig(vocab_size=d["vocab_size"], hidden_size=d["hidden_size"], intermediate_size=d["intermediate_size"],
                      num_hidden_layers=d["num_layers"], num_attention_heads=d["num_heads"],
                      num_key_value_heads=d["num_kv_heads"], max_position_embeddings=max(d["max_position"], 2048),
                      tie_word_embeddings=False)
    torch.manual_seed(1234)
    model = optimize_model(LlamaForCausalLM(cfg).eval(), low_bit=args.low_bit)
    model.generation_config.eos_token_id = None
    model.generation_config.pad_token_id = 0
    par = args.parallelism if world > 1 else "dp"
    if par == "tp":
        from bigdl_amd.parallel.tensor_parallel import tensor_parallel
        model = tensor_parallel(model)
    elif par == "pp":
        from bigdl_amd.transformers.pipeline_parallel import pipeline_parallel
        model = pipeline_parallel(model, world, torch.float32)
    g = torch.Generator().manual_seed(0 if par in ("tp", "pp") else rank)
    prompt = torch.randint(3, cfg.vocab_size, (1, args.in_len), generator=g)

    def trial():
        times = []
        t0 = time.perf_counter()
        if par == "pp":
            out = model.generate(prompt, max_new_tokens=args.out_len)
            return model.first_token_time, model.rest_cost_mean, 0.0
        from transformers import StoppingCriteria, StoppingCriteriaList

        class Clock(StoppingCriteria):          # one tick per generated token
            def __call__(self, ids, scores, **kw):
                times.append(time.perf_counter())
                return False
        out = model._hf_generate(prompt, max_new_tokens=args.out_len, min_new_tokens=args.out_len, do_sample=False,
                                 stopping_criteria=StoppingCriteriaList([Clock()])) if hasattr(model, "_hf_generate") else \
            model.generate(prompt, max_new_tokens=args.out_len, min_new_tokens=args.out_len, do_sample=False,
                           stopping_criteria=StoppingCriteriaList([Clock()]))
        assert out.shape[1] == args.in_len + args.out_len
        first = times[0] - t0
        rest = (times[-1] - times[0]) / max(1, len(times) - 1)
        return first, rest, 0.0

    stats = timed_trials(args, trial, dist, torch.device("cpu"))
    if rank == 0:
        report(args, world, par, "test-size llama (random-init)" if args.model == "tiny" else model_config(args.model)[1],
               stats, extra={"device": "cpu (gloo)"})


def qlora(args, rank, world, dist):
    """QLoRA training steps (data parallel): delegates to benchmark/qlora_bench.py in this process group"""
    sys.argv = [sys.argv[0], "--steps", str(args.steps), "--warmup", str(args.warmup),
                "--model", "llama2-7b" if args.model == "llama2-7b" else "tiny", "--full-targets"] + \
               ([] if args.no_graph else ["--graph"])
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "benchmark"))
    import qlora_bench
    qlora_bench.main(process_group_ready=dist is not None)


if __name__ == "__main__":
    main()
