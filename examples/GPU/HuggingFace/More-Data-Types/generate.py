"""Every low-bit format on the same prompt: load time, memory and next-token latency side by side
(reference: python/llm/example/GPU/HuggingFace/More-Data-Types/transformers_low_bit_pipeline.py)."""
import argparse
import time

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

FORMATS = ["sym_int4", "asym_int4", "sym_int5", "asym_int5", "sym_int8", "nf4", "nf3", "fp4", "fp6", "fp8_e4m3", "fp8_e5m2",
           "q4_k", "q6_k", "fp16", "bf16"]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", nargs="+", default=["sym_int4", "nf4", "fp6", "fp8_e5m2"], choices=FORMATS)
    ap.add_argument("--prompt", default="Once upon a time, there existed a little girl who liked to have adventures.")
    ap.add_argument("--n-predict", type=int, default=32)
    a = ap.parse_args()
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    for fmt in a.low_bit:
        t0 = time.time()
        model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=fmt, trust_remote_code=True)
        model = model.half().to("cuda")
        load = time.time() - t0
        with torch.inference_mode():
            model.generate(ids, max_new_tokens=8)
            out = model.generate(ids, max_new_tokens=a.n_predict)
        print(f"[{fmt}] load {load:.1f} s, {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB, "
              f"{getattr(model, 'rest_cost_mean', 0) * 1e3:.2f} ms/token: {tok.decode(out[0], skip_special_tokens=True)!r}")
        del model
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
