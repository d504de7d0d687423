"""Self-speculative decoding: fp16 target + sym_int4 draft of the same weights
(reference: python/llm/example/GPU/Speculative-Decoding/Self-Speculation/llama2/speculative.py)."""
import argparse

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="Once upon a time")
    ap.add_argument("--n-predict", type=int, default=128)
    ap.add_argument("--th-stop-draft", type=float, default=0.6)
    a = ap.parse_args()
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, optimize_model=True, torch_dtype=torch.float16,
                                                 load_in_low_bit="fp16", speculative=True, use_cache=True)
    model = model.to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    out = model.generate(ids, max_new_tokens=a.n_predict, th_stop_draft=a.th_stop_draft, do_sample=False)
    print(tok.decode(out[0], skip_special_tokens=True))
    print(f"first token {model.first_token_time:.3f}s; {model.n_token_generated} tokens in {model.e2e_time_without_first:.3f}s; "
          f"drafted {model.n_drafted}, accepted {model.n_matched}")
