"""Single-GPU serving with iteration-level batching on the native engine: up to ``--max-num-seqs`` requests share the
slots of one KV cache and advance together with one hipGraph replay per token (bigdl_amd/serving/continuous.py).
OpenAI / TGI routes as in the lightweight server.  The scheduler lets the running sequences ride on the prompt pass of
a new request, admits long prompts in chunks and packs waiting short prompts into one pass.  ``--kv-cache int4`` (or
``fp8``) selects a quantized KV cache (``IPEX_LLM_QUANTIZE_KV_CACHE``)."""
import argparse
import os

import uvicorn

from bigdl_amd.serving.fastapi import FastApp
from bigdl_amd.serving.fastapi.model_worker import BatchedModelWorker

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--max-num-seqs", type=int, default=32)
    ap.add_argument("--max-seq-len", type=int, default=4096)
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--kv-cache", default=None, choices=["fp16", "fp8", "int4"])
    a = ap.parse_args()
    if a.kv_cache is not None:
        os.environ["IPEX_LLM_QUANTIZE_KV_CACHE"] = {"fp16": "0", "fp8": "1", "int4": "int4"}[a.kv_cache]
    worker = BatchedModelWorker(a.repo_id_or_model_path, a.low_bit, max_num_seqs=a.max_num_seqs, max_seq_len=a.max_seq_len)
    uvicorn.run(FastApp(worker=worker, served_model_name=a.repo_id_or_model_path).app, host="0.0.0.0", port=a.port)
