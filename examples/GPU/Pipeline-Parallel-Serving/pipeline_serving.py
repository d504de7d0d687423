"""Pipeline-parallel serving: every rank holds a slice of the layers, rank 0 runs the HTTP server, up to
``world_size`` micro-batches are in flight (reference: python/llm/example/GPU/Pipeline-Parallel-Serving/pipeline_serving.py).
    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 4 pipeline_serving.py --repo-id-or-model-path <path>"""
import argparse
import asyncio

import torch
import torch.distributed as dist
from transformers import AutoTokenizer

from ipex_llm.transformers import init_pipeline_parallel
from ipex_llm.transformers.pipeline_parallel import PPModelWorker

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--max-num-seqs", type=int, default=8)
    ap.add_argument("--max-prefilled-seqs", type=int, default=0)
    a = ap.parse_args()
    init_pipeline_parallel()
    rank, world = dist.get_rank(), dist.get_world_size()
    for r in range(world):                       # load one rank at a time: bounds the host memory used for conversion
        if r == rank:
            worker = PPModelWorker(a.repo_id_or_model_path, rank, world, a.low_bit, a.max_num_seqs, a.max_prefilled_seqs,
                                   torch_dtype=torch.float16)
        dist.barrier()
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    if rank == 0:
        import uvicorn
        from ipex_llm.serving.fastapi import FastApp
        uvicorn.run(FastApp(worker=worker).app, host="0.0.0.0", port=a.port)
    else:
        async def loop():
            while True:
                await worker.process_step()
        asyncio.run(loop())
