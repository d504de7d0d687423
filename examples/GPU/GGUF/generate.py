"""Load a GGUF file directly (reference: python/llm/example/GPU/HuggingFace/Advanced-Quantizations/GGUF/generate.py)."""
import argparse

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True, help="path to a .gguf file")
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    a = ap.parse_args()
    model, tok = AutoModelForCausalLM.from_gguf(a.model, low_bit=a.low_bit)
    model = model.half().to("cuda")
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    print(tok.decode(model.generate(ids, max_new_tokens=a.n_predict)[0], skip_special_tokens=True))
