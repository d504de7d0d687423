"""save_low_bit / load_low_bit round trip (reference: python/llm/example/GPU/HuggingFace/Save-Load/generate.py)."""
import argparse

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--save-path", required=True)
    ap.add_argument("--low-bit", default="sym_int4")
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit)
    model.save_low_bit(a.save_path)
    AutoTokenizer.from_pretrained(a.repo_id_or_model_path).save_pretrained(a.save_path)
    model = AutoModelForCausalLM.load_low_bit(a.save_path).half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.save_path)
    ids = tok("What is AI?", return_tensors="pt").input_ids.to("cuda")
    print(tok.decode(model.generate(ids, max_new_tokens=32)[0], skip_special_tokens=True))
