"""Batched tensor-parallel serving: every rank holds a shard of the native engine and the same continuous-batching
scheduler; rank 0 runs the HTTP server and broadcasts what it admits each step as fixed-shape tensors
(bigdl_amd/serving/tp.py).  Reference: python/llm/example/GPU/Deepspeed-AutoTP-FastAPI/serving.py (rank 0 batches up
to ``max_num_seqs`` / ``max_num_batched_tokens`` and broadcasts the batch).

    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 2 serving.py --repo-id-or-model-path <path>
"""
import argparse

import torch
from transformers import AutoTokenizer

from bigdl_amd.models.hf_patch import get_native_engine, native_engine_supported
from bigdl_amd.parallel import init_distributed
from bigdl_amd.parallel.comm import staggered_wait
from bigdl_amd.parallel.tensor_parallel import shard_engine
from bigdl_amd.serving.tp import TensorParallelBatcher
from bigdl_amd.transformers.utils import engine_kv_mode
from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--max-num-seqs", type=int, default=16)
    ap.add_argument("--max-num-batched-tokens", type=int, default=8192)
    ap.add_argument("--max-seq-len", type=int, default=4096)
    a = ap.parse_args()
    rank, world, device = init_distributed()
    staggered_wait()                                         # RANK_WAIT_TIME=<s>: ranks load one after another
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, torch_dtype=torch.float16)
    model = model.half().to(device).eval()
    if not native_engine_supported(model):
        raise SystemExit("this model family has no native engine: see generate.py in this directory for the module path")
    engine = shard_engine(get_native_engine(model), rank, world)
    tp = TensorParallelBatcher(engine, max_num_seqs=a.max_num_seqs, max_seq_len=a.max_seq_len, kv_fp8=engine_kv_mode(),
                               max_batched_tokens=a.max_num_batched_tokens)
    if rank == 0:
        import uvicorn
        from bigdl_amd.serving.fastapi import FastApp
        from bigdl_amd.serving.fastapi.model_worker import BatchedModelWorker
        tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
        worker = BatchedModelWorker(model=model, tokenizer=tok, device=str(device), batcher=tp)
        uvicorn.run(FastApp(worker=worker, served_model_name=a.repo_id_or_model_path).app, host="0.0.0.0", port=a.port)
        tp.shutdown()
    else:
        tp.follow()
