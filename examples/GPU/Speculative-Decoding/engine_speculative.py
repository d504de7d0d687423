"""Self-speculative decoding statistics on the native engine: the fp16 model verifies, its sym_int4 conversion drafts on
the SAME KV cache (reference: python/llm/example/GPU/Speculative-Decoding/<model>/speculative.py, which prints the
first-token / per-token latency of ``speculative=True`` models).

    python engine_speculative.py --repo-id-or-model-path <path> --prompt "..." [--max-draft 6]
"""
import argparse

import torch
from transformers import AutoTokenizer

from ipex_llm.transformers import AutoModelForCausalLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="Explain how a transformer decoder generates text, step by step.")
    ap.add_argument("--n-predict", type=int, default=128)
    ap.add_argument("--max-draft", type=int, default=6)
    ap.add_argument("--draft-layers", default=None,
                    help="the int4 draft runs only this many of its layers (or the comma-separated layer indices given) on the "
                         "target's cache: a cheaper draft; how many layers a model lets a draft skip is found by trying")
    a = ap.parse_args()
    # speculative=True: fp16 target + sym_int4 draft of the same checkpoint, generate() routes to the speculative loop
    model = AutoModelForCausalLM.from_pretrained(a.repo_id_or_model_path, load_in_low_bit="fp16", optimize_model=True,
                                                 torch_dtype=torch.float16, speculative=True, use_cache=True)
    model = model.to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path)
    ids = tok(a.prompt, return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        for _ in range(2):
            out = model.generate(ids, max_new_tokens=a.n_predict, max_step_draft=a.max_draft, do_sample=False,
                                 **({"draft_layers": a.draft_layers} if a.draft_layers else {}))
    print(tok.decode(out[0, ids.shape[1]:], skip_special_tokens=True))
    n = model.n_token_generated
    print(f"-- first token {model.first_token_time * 1e3:.1f} ms; {n - 1} more tokens in {model.e2e_time_without_first:.3f} s "
          f"({model.e2e_time_without_first / max(1, n - 1) * 1e3:.2f} ms/token)")
    print(f"-- drafted {model.n_drafted}, accepted {model.n_matched} ({100 * model.n_matched / max(1, model.n_drafted):.0f} %), "
          f"{len(model.accept_num)} rounds, {sum(model.accept_num) / max(1, len(model.accept_num)):.2f} tokens per round")
