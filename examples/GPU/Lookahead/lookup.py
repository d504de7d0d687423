"""Prompt-lookup decoding (reference: python/llm/example/GPU/Lookahead/; ``lookahead`` generate kwarg)."""
import argparse

from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt-file", required=True, help="a long document followed by a summarisation instruction")
    ap.add_argument("--n-predict", type=int, default=128)
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_4bit=True).half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(open(a.prompt_file).read(), return_tensors="pt").input_ids.to("cuda")
    out = model.generate(ids, lookahead=3, max_matching_ngram_size=2, max_new_tokens=a.n_predict, do_sample=False)
    print(tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True))
    print(f"{model.n_token_generated} tokens, {model.rest_cost_mean * 1e3:.2f} ms/token, "
          f"{model.n_matched}/{model.n_drafted} lookup tokens accepted")
