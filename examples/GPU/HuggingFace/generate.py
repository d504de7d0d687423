"""Low-bit generation on one MI355X (reference: python/llm/example/GPU/HuggingFace/LLM/llama2/generate.py)."""
import argparse
import time

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, optimize_model=True,
                                                 trust_remote_code=True, use_cache=True)
    model = model.half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        model.generate(ids, max_new_tokens=a.n_predict)                 # warm-up: builds the decode graph
        t0 = time.time()
        out = model.generate(ids, max_new_tokens=a.n_predict)
        torch.cuda.synchronize()
    print(f"Inference time: {time.time() - t0:.3f} s  (first token {model.first_cost * 1e3:.1f} ms, "
          f"next tokens {model.rest_cost_mean * 1e3:.2f} ms/token)")
    print(tok.decode(out[0], skip_special_tokens=True))
