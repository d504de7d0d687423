"""CPU inference through the C++ quantized matmul (reference: python/llm/example/CPU/HuggingFace-Models)."""
import argparse

from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--n-predict", type=int, default=32)
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True)
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(a.prompt, return_tensors="pt").input_ids
    print(tok.decode(model.generate(ids, max_new_tokens=a.n_predict)[0], skip_special_tokens=True))
