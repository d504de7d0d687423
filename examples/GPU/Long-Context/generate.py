"""Long prompts on one GPU: fp8 KV cache and SnapKV-style KV compression
(reference: python/llm/example/GPU/Long-Context/{LLaMA2-32K,Chatglm3-32K}/generate.py; the reference tops out at 7-12 K
input tokens on a 16 GB Arc A770 -- 288 GB of HBM take the same model to its full context window)."""
import argparse
import os

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt-file", required=True)
    ap.add_argument("--n-predict", type=int, default=128)
    ap.add_argument("--fp8-kv", action="store_true", help="IPEX_LLM_QUANTIZE_KV_CACHE=1: e5m2 KV cache, half the bytes")
    ap.add_argument("--compress-kv", action="store_true", help="IPEX_LLM_COMPRESS_KV_CACHE=1: keep 1024 prompt entries")
    a = ap.parse_args()
    if a.fp8_kv:
        os.environ["IPEX_LLM_QUANTIZE_KV_CACHE"] = "1"
    if a.compress_kv:
        os.environ["IPEX_LLM_COMPRESS_KV_CACHE"] = "1"
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True, trust_remote_code=True)
    model = model.half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    ids = tok(open(a.prompt_file).read(), return_tensors="pt").input_ids.to("cuda")
    print(f"prompt length: {ids.shape[1]} tokens")
    with torch.inference_mode():
        out = model.generate(ids, max_new_tokens=a.n_predict)
    print(tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True))
    print(f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB")
