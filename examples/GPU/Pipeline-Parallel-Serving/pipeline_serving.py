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
    ap.add_argument("--max-seq-len", type=int, default=4096)
    ap.add_argument("--engine", action=argparse.BooleanOptionalAction, default=True,
                    help="run the stages on the native engine (default); --no-engine: HuggingFace modules")
    a = ap.parse_args()
    ap_engine = a.engine
    init_pipeline_parallel()
    rank, world = dist.get_rank(), dist.get_world_size()
    for r in range(world):                       # load one rank at a time: bounds the host memory used for conversion
        if r == rank:
            if ap_engine:
                # every stage step on the native engine: one graph replay per micro-batch step, packed prompt passes,
                # partial prefill in pieces of --max-prefilled-seqs (bigdl_amd/parallel/pipeline.py::EnginePPWorker)
                from bigdl_amd.models.hf_patch import get_native_engine, native_engine_supported
                from bigdl_amd.parallel.pipeline import EnginePPWorker, split_engine
                from ipex_llm.transformers import AutoModelForCausalLM
                m = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit,
                                                         torch_dtype=torch.float16)
                if not native_engine_supported(m):
                    raise SystemExit("this model family has no native engine: run without --engine")
                stage = split_engine(get_native_engine(m.half().to(f"cuda:{torch.cuda.current_device()}")), rank, world)
                worker = EnginePPWorker(stage, rank, world, a.max_num_seqs, a.max_prefilled_seqs,
                                        tokenizer=AutoTokenizer.from_pretrained(a.repo_id_or_model_path),
                                        max_seq_len=a.max_seq_len)
            else:
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
