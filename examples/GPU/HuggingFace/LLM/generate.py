"""One generation script for every model family the reference ships a directory for
(reference: python/llm/example/GPU/HuggingFace/LLM/<family>/generate.py -- 50-odd copies that differ in the default
checkpoint and the prompt format).  The prompt formats live in ``bigdl_amd.utils.prompts``; README.md next to this file
lists the command line per family.

    python generate.py --family llama2 --repo-id-or-model-path meta-llama/Llama-2-7b-chat-hf --prompt "What is AI?"
    python generate.py --family qwen2 --repo-id-or-model-path Qwen/Qwen2-7B-Instruct --low-bit fp8_e5m2 --n-predict 64
"""
import argparse
import time

import torch
from transformers import AutoTokenizer

from bigdl_amd.utils.prompts import build_prompt, families, family_of, uses_chat_template
from ipex_llm.transformers import AutoModel, AutoModelForCausalLM, AutoModelForSeq2SeqLM

# families whose checkpoints load through another auto class
AUTO_CLASS = {"chatglm2": AutoModel, "chatglm3": AutoModel, "codegeex2": AutoModel, "glm4": AutoModelForCausalLM,
              "flan-t5": AutoModelForSeq2SeqLM}

if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Predict tokens with generate() for any supported model family")
    ap.add_argument("--family", choices=list(families()), default=None,
                    help="prompt format to use; default: derived from the checkpoint's config, else the tokenizer's chat "
                         "template, else the prompt as typed")
    ap.add_argument("--repo-id-or-model-path", required=True)
    ap.add_argument("--prompt", default="What is AI?")
    ap.add_argument("--system", default="")
    ap.add_argument("--n-predict", type=int, default=32)
    ap.add_argument("--low-bit", default="sym_int4")
    ap.add_argument("--modelscope", action="store_true", help="resolve the id on the ModelScope hub")
    a = ap.parse_args()

    cls = AUTO_CLASS.get(a.family or "", AutoModelForCausalLM)
    model = cls.from_pretrained(a.repo_id_or_model_path, load_in_low_bit=a.low_bit, optimize_model=True,
                                trust_remote_code=True, use_cache=True, model_hub="modelscope" if a.modelscope else "huggingface")
    model = model.half().to("cuda")
    tok = AutoTokenizer.from_pretrained(a.repo_id_or_model_path, trust_remote_code=True)

    family = a.family or family_of(model.config)
    if family is not None:
        text = build_prompt(family, a.prompt, system=a.system, tokenizer=tok if uses_chat_template(family) else None)
    elif getattr(tok, "chat_template", None):
        msgs = ([{"role": "system", "content": a.system}] if a.system else []) + [{"role": "user", "content": a.prompt}]
        text = tok.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
    else:
        text = a.prompt
    ids = tok(text, return_tensors="pt").input_ids.to("cuda")
    with torch.inference_mode():
        model.generate(ids, max_new_tokens=a.n_predict)                 # warm-up: engine build + decode graph capture
        t0 = time.time()
        out = model.generate(ids, max_new_tokens=a.n_predict)
        torch.cuda.synchronize()
    print(f"Inference time: {time.time() - t0:.3f} s")
    if hasattr(model, "first_cost"):
        print(f"first token {model.first_cost * 1e3:.1f} ms, next tokens {model.rest_cost_mean * 1e3:.2f} ms/token")
    print("-" * 20, "Prompt", "-" * 20)
    print(text)
    print("-" * 20, "Output", "-" * 20)
    print(tok.decode(out[0], skip_special_tokens=True))
