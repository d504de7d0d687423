"""Aggregate decode throughput of one GPU under continuous batching (native engine, hipGraph step over all slots).

    python benchmark/serving_throughput.py --slots 1,8,32,64 --in-len 1024 --out-len 128

Random-init Llama-2-7B sym_int4 weights, synthetic prompts (no network for checkpoints); prints one JSON line per
slot count: requests/s, generated tokens/s, mean time to first token, mean time per output token.

``--tp N``: the same load on a tensor-parallel group of N ranks (``shard_engine`` + ``serving/tp.py``): under torchrun
one rank per GPU over RCCL; started plainly, the script spawns N ranks itself that share GPU 0 over gloo (a rehearsal of
the path on a one-GPU box, not a speed measurement)."""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models import decoder as D
from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
from bigdl_amd.serving.continuous import ContinuousBatcher, SeqRequest


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", default="1,8,32,64")
    ap.add_argument("--in-len", type=int, default=1024)
    ap.add_argument("--out-len", type=int, default=128)
    ap.add_argument("--requests-per-slot", type=int, default=2)
    ap.add_argument("--model", default="llama2-7b")
    ap.add_argument("--kv", default="fp8")
    ap.add_argument("--prefill-per-step", type=int, default=8, help="prompts admitted per step (packed into one pass)")
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel ranks")
    args = ap.parse_args()
    if args.tp > 1:
        return tp_main(args)
    cfg = DecoderConfig(**getattr(D, {"llama2-7b": "LLAMA2_7B", "tinyllama": "TINY_LLAMA"}.get(args.model, args.model)))
    # --model also takes the name of any shape constant of models/decoder.py (LLAMA3_8B, QWEN2_7B, ...)
    eng = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=1)
    g = torch.Generator().manual_seed(0)
    for slots in [int(s) for s in args.slots.split(",")]:
        n_req = slots * args.requests_per_slot
        cb = ContinuousBatcher(eng, max_num_seqs=slots, max_seq_len=args.in_len + args.out_len + 1, kv_fp8={"fp8": True, "fp16": False, "int4": "int4"}[args.kv],
                               max_prefill_per_step=args.prefill_per_step)
        # warm-up: graph capture + one short request
        cb.submit(SeqRequest("w", torch.randint(3, cfg.vocab_size, (64,), generator=g).tolist(), max_new_tokens=4))
        cb.run_until_done()
        reqs = [cb.submit(SeqRequest(str(i), torch.randint(3, cfg.vocab_size, (args.in_len,), generator=g).tolist(),
                                     max_new_tokens=args.out_len)) for i in range(n_req)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in reqs:
            r.arrival = t0
        cb.run_until_done()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gen = sum(len(r.output_ids) for r in reqs)
        tpot = [(r.finish_time - r.first_token_time) / (len(r.output_ids) - 1) for r in reqs]
        print(json.dumps({"bench": "continuous_batching", "model": args.model, "slots": slots, "requests": n_req,
                          "in_len": args.in_len, "out_len": args.out_len, "wall_s": round(dt, 3),
                          "gen_tokens_per_s": round(gen / dt, 1), "requests_per_s": round(n_req / dt, 2),
                          "mean_tpot_ms": round(1e3 * sum(tpot) / len(tpot), 3),
                          "peak_mem_gb": round(torch.cuda.max_memory_reserved() / 2 ** 30, 2), "steps": cb.steps,
                          "mixed_steps": cb.mixed_steps, "packed_prompts": cb.packed_prompts,
                          "packed_mixed_steps": getattr(cb, "packed_mixed_steps", 0)}), flush=True)


def tp_main(args):
    import torch.distributed as dist
    if "RANK" not in os.environ:                 # not under torchrun: N ranks on GPU 0, gloo
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = [subprocess.Popen([sys.executable] + sys.argv,
                                  env=dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(args.tp), MASTER_ADDR="127.0.0.1",
                                           MASTER_PORT=str(port), BIGDL_TP_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
                 for r in range(args.tp)]
        sys.exit(max(p.wait() for p in procs))
    shared = os.environ.get("BIGDL_TP_SHARED_GPU") == "1"
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0 if shared else int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("gloo" if shared else "nccl")
    from bigdl_amd.parallel.tensor_parallel import shard_engine
    from bigdl_amd.serving.tp import TensorParallelBatcher
    cfg = DecoderConfig(**getattr(D, {"llama2-7b": "LLAMA2_7B", "tinyllama": "TINY_LLAMA"}.get(args.model, args.model)))
    eng = shard_engine(build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=1), rank, args.tp)
    g = torch.Generator().manual_seed(0)
    for slots in [int(s) for s in args.slots.split(",")]:
        n_req = slots * args.requests_per_slot
        tp = TensorParallelBatcher(eng, max_num_seqs=slots, max_seq_len=args.in_len + args.out_len + 1,
                                   kv_fp8={"fp8": True, "fp16": False, "int4": "int4"}[args.kv],
                                   max_prefill_per_step=args.prefill_per_step, max_admit_per_step=args.prefill_per_step,
                                   max_batched_tokens=max(8192, args.in_len))
        if rank != 0:
            tp.follow()
            continue
        tp.submit(SeqRequest("w", torch.randint(3, cfg.vocab_size, (64,), generator=g).tolist(), max_new_tokens=4))
        while tp.has_work():
            tp.step()
        reqs = [tp.submit(SeqRequest(str(i), torch.randint(3, cfg.vocab_size, (args.in_len,), generator=g).tolist(),
                                     max_new_tokens=args.out_len)) for i in range(n_req)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in reqs:
            r.arrival = t0
        while tp.has_work():
            tp.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tp.shutdown()
        gen = sum(len(r.output_ids) for r in reqs)
        print(json.dumps({"bench": "tp_continuous_batching", "tp": args.tp, "shared_gpu": shared, "model": args.model,
                          "slots": slots, "requests": n_req, "in_len": args.in_len, "out_len": args.out_len,
                          "wall_s": round(dt, 3), "gen_tokens_per_s": round(gen / dt, 1), "steps": tp.steps,
                          "payload_steps": tp.payload_steps,
                          "graph_decode": bool(eng.use_graph and eng._reduce_capturable())}), flush=True)
    dist.barrier()
    if eng.tp_fast_allreduce is not None:
        eng.tp_fast_allreduce.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
