"""``optimize_model`` on a plain PyTorch / HuggingFace model object, then ``save_low_bit`` and the low-memory reload
(reference: python/llm/example/GPU/PyTorch-Models/Model/llama2/generate.py, docs/mddocs/PythonAPI/optimize.md:6-77)."""
import argparse

import torch
from transformers import AutoModelForCausalLM, AutoTokenizer

from ipex_llm import optimize_model
from ipex_llm.optimize import load_low_bit, low_memory_init

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--save-dir", default="./optimized")
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, torch_dtype="auto", low_cpu_mem_usage=True)
    model = optimize_model(model, low_bit=a.low_bit)
    model.save_low_bit(a.save_dir)
    with low_memory_init():                      # meta-device construction: no full-precision weights are materialised
        fresh = AutoModelForCausalLM.from_pretrained(a.save_dir, torch_dtype="auto")
    fresh = load_low_bit(fresh, a.save_dir).half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok("What is AI?", return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        print(tok.decode(fresh.generate(ids, max_new_tokens=32)[0], skip_special_tokens=True))
