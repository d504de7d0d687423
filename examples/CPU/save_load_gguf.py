"""Host-only workflow: convert once, save the low-bit form, reload it instantly, or load a GGUF file directly
(reference: python/llm/example/CPU/HF-Transformers-AutoModels/Save-Load and Advanced-Quantizations/GGUF).

    python save_load_gguf.py --repo-id-or-model-path <hf dir> --save-dir ./llama-int4
    python save_load_gguf.py --gguf model.Q4_0.gguf
"""
import argparse
import time

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path")
    ap.add_argument("--save-dir")
    ap.add_argument("--gguf")
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--n-predict", type=int, default=32)
    a = ap.parse_args()
    if a.gguf:
        model, tok = AutoModelForCausalLM.from_gguf(a.gguf)
    else:
        t0 = time.perf_counter()
        model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit)
        print(f"converted in {time.perf_counter() - t0:.1f} s")
        tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
        if a.save_dir:
            model.save_low_bit(a.save_dir)
            tok.save_pretrained(a.save_dir)
            t0 = time.perf_counter()
            model = AutoModelForCausalLM.load_low_bit(a.save_dir)
            print(f"reloaded the low-bit checkpoint in {time.perf_counter() - t0:.1f} s")
    ids = tok(a.prompt, return_tensors="pt").input_ids
    with torch.inference_mode():
        out = model.generate(ids, max_new_tokens=a.n_predict, do_sample=False)
    print(tok.decode(out[0], skip_special_tokens=True))
