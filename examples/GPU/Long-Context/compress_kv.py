"""SnapKV prompt compression on a long prompt (reference: python/llm/example/GPU/Long-Context, IPEX_LLM_COMPRESS_KV_CACHE=1):
after the prompt pass every layer keeps the 1024 keys its last 32 queries attend to most; decode steps read 1024 + n keys
instead of the whole prompt while positions keep counting from the real length.

    python compress_kv.py --repo-id-or-model-path <path> --prompt-file long.txt
"""
import argparse
import os
import time

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt-file", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--n-predict", type=int, default=128)
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, optimize_model=True,
                                                 use_cache=True).half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(open(a.prompt_file).read(), return_tensors="pt").input_ids.to("cuda")
    for flag in ("0", "1"):
        os.environ["IPEX_LLM_COMPRESS_KV_CACHE"] = flag
        with torch.inference_mode():
            model.generate(ids, max_new_tokens=4)
            t0 = time.perf_counter()
            out = model.generate(ids, max_new_tokens=a.n_predict, do_sample=False)
            torch.cuda.synchronize()
        print(f"compress={flag}: prompt {ids.shape[1]} tokens, first token {model.first_cost * 1e3:.1f} ms, "
              f"{model.rest_cost_mean * 1e3:.2f} ms/token, total {time.perf_counter() - t0:.2f} s")
        print("   ", tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True)[:200].replace("\n", " "))
